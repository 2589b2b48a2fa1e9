// fused mul_accumulate-chain kernel instantiations: u32
#define INST_T uint32_t
#include "ntt_ext_inst.inc"
