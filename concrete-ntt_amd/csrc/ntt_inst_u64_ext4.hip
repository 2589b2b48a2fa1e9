// fused mul_accumulate-chain kernel instantiations: u64, 4 outputs
#define INST_T uint64_t
#define INST_NOUT 4
#include "ntt_ext_inst.inc"
