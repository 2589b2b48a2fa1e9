// fused mul_accumulate-chain kernel instantiations: u32, 2 outputs
#define INST_T uint32_t
#define INST_NOUT 2
#include "ntt_ext_inst.inc"
