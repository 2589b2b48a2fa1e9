// fused product kernel instantiations: u32
#define INST_T uint32_t
#include "ntt_mul_inst.inc"
