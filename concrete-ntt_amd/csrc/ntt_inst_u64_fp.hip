// CLS_FP kernel instantiations (64-bit words, p < 2^50, residues held as doubles): forward, inverse and the fused
// product, every LDS-resident size.
#define INST_FPCLS CLS_FP
#include "ntt_fp_inst.inc"
