// fused mul_accumulate-chain kernel instantiations: u64
#define INST_T uint64_t
#include "ntt_ext_inst.inc"
