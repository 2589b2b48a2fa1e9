// NTT kernel instantiations: u64, fwd
#define INST_T uint64_t
#define INST_INV false
#include "ntt_inst.inc"
