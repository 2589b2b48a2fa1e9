// NTT kernel instantiations: u64, inv
#define INST_T uint64_t
#define INST_INV true
#include "ntt_inst.inc"
