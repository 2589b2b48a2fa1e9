// fused mul_accumulate-chain kernel instantiations: u64, 1 output
#define INST_T uint64_t
#define INST_NOUT 1
#include "ntt_ext_inst.inc"
