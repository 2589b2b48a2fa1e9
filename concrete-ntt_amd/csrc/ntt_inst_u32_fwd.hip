// NTT kernel instantiations: u32, fwd
#define INST_T uint32_t
#define INST_INV false
#include "ntt_inst.inc"
