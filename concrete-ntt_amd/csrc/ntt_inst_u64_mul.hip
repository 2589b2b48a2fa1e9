// fused product kernel instantiations: u64
#define INST_T uint64_t
#include "ntt_mul_inst.inc"
