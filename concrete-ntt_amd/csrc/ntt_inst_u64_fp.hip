// CLS_FP kernel instantiations (64-bit words, p < 2^50, residues held as doubles): forward, inverse and the fused
// product, every LDS-resident size.
#include "ntt_launch_one.hpp"
#include "ntt_mul_one.hpp"

namespace cntt {

template <bool INV, int LOGN>
static hipError_t fp_logn(int logn, uint64_t *data, const TwPair<uint64_t> *tw, const ModParams<uint64_t> &P, uint32_t nsub,
                          hipStream_t stream) {
    if constexpr (LOGN > MaxLdsLogN<uint64_t>::value) {
        return hipErrorInvalidValue;
    } else {
        if (logn == LOGN) return launch_one<uint64_t, LOGN, INV, CLS_FP, false>(data, tw, P, nsub, 0, stream);
        return fp_logn<INV, LOGN + 1>(logn, data, tw, P, nsub, stream);
    }
}

hipError_t launch_ntt_fp(int logn, bool inv, uint64_t *data, const TwPair<uint64_t> *tw, const ModParams<uint64_t> &P,
                         uint32_t nsub, hipStream_t stream) {
    return inv ? fp_logn<true, MinLogN<uint64_t>::value>(logn, data, tw, P, nsub, stream)
               : fp_logn<false, MinLogN<uint64_t>::value>(logn, data, tw, P, nsub, stream);
}

template <int LOGN>
static hipError_t fp_mul_logn(int logn, uint64_t *lhs, const uint64_t *rhs, const TwPair<uint64_t> *twf,
                              const TwPair<uint64_t> *twi, const ModParams<uint64_t> &P, uint32_t nsub, hipStream_t stream) {
    if constexpr (LOGN > 12) {
        return hipErrorNotSupported;
    } else {
        using K0 = NttKernel<uint64_t, LOGN, false, CLS_LAZY, false>;
        if (logn == LOGN) {
            if constexpr (K0::NPASS > 1 && K0::TPP <= 256 && (size_t)K0::IMG_ENTRIES * sizeof(TwPair<uint64_t>) <= 32768)
                return mul_one<uint64_t, LOGN, CLS_FP>(lhs, rhs, twf, twi, P, nsub, stream);
            else
                return hipErrorNotSupported;
        }
        return fp_mul_logn<LOGN + 1>(logn, lhs, rhs, twf, twi, P, nsub, stream);
    }
}

hipError_t launch_mul_ntt_fp(int logn, uint64_t *lhs, const uint64_t *rhs_ntt, const TwPair<uint64_t> *twf,
                             const TwPair<uint64_t> *twi, const ModParams<uint64_t> &P, uint32_t nsub, hipStream_t stream) {
    return fp_mul_logn<MinLogN<uint64_t>::value>(logn, lhs, rhs_ntt, twf, twi, P, nsub, stream);
}

}  // namespace cntt
