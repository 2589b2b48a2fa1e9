// product::Plan u32x2 fused forward / inverse kernels (product_fused.hpp): instantiations and dispatch.
#include "product_fused.hpp"
#include "ntt_launch.hpp"

namespace cntt {

template <int LOGN, int CLS>
static hipError_t pf_one(bool inv, uint64_t *standard, uint32_t *res32, const ProductFusedTables &F, const ProductArgs &A,
                         uint32_t batch, bool flag, hipStream_t st) {
    using K0 = NttKernel<uint32_t, LOGN, false, CLS_LAZY, false, 1>;
    if constexpr (K0::NPASS > 1 && K0::TPP <= 256) {
        constexpr int BLK = 256, PPB = BLK / K0::TPP;
        const uint32_t grid = (batch + PPB - 1) / PPB;
        if (inv)
            hipLaunchKernelGGL((product_inv2_kernel<LOGN, CLS, BLK>), dim3(grid), dim3(BLK), 0, st, standard, res32, F, A, batch,
                               flag ? 1u : 0u);
        else
            hipLaunchKernelGGL((product_fwd2_kernel<LOGN, CLS, BLK>), dim3(grid), dim3(BLK), 0, st, res32,
                               (const uint64_t *)standard, F, A, batch, flag ? 1u : 0u);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int LOGN>
static hipError_t pf_logn(int logn, int cls, bool inv, uint64_t *standard, uint32_t *res32, const ProductFusedTables &F,
                          const ProductArgs &A, uint32_t batch, bool flag, hipStream_t st) {
    if constexpr (LOGN > 12) {
        return hipErrorNotSupported;
    } else {
        if (logn == LOGN) {
            switch (cls) {
            case CLS_LAZY: return pf_one<LOGN, CLS_LAZY>(inv, standard, res32, F, A, batch, flag, st);
            case CLS_STRICT: return pf_one<LOGN, CLS_STRICT>(inv, standard, res32, F, A, batch, flag, st);
            case CLS_FPW: return pf_one<LOGN, CLS_FPW>(inv, standard, res32, F, A, batch, flag, st);
            default: return pf_one<LOGN, CLS_GENERIC>(inv, standard, res32, F, A, batch, flag, st);
            }
        }
        return pf_logn<LOGN + 1>(logn, cls, inv, standard, res32, F, A, batch, flag, st);
    }
}

hipError_t launch_product_fused2(int logn, int cls, bool inv, uint64_t *standard, uint32_t *res32, const void *tables,
                                 const ProductArgs &A, uint32_t batch, bool flag, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    if (logn < 5 || logn > 12) return hipErrorNotSupported;
    return pf_logn<5>(logn, cls, inv, standard, res32, *static_cast<const ProductFusedTables *>(tables), A, batch, flag, st);
}

}  // namespace cntt
