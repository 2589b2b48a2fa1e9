// CLS_PM64 kernel instantiations (64-bit words, p = 2^64 - c with c < 2^32: Solinas and its neighbours): forward,
// inverse and the fused product, every LDS-resident size.
#define INST_FPCLS CLS_PM64
#include "ntt_fp_inst.inc"
