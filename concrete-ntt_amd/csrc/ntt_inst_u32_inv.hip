// NTT kernel instantiations: u32, inv
#define INST_T uint32_t
#define INST_INV true
#include "ntt_inst.inc"
