// whole-polymul kernel instantiations: native kind 5
#define INST_KIND 5
#include "native_fused_inst.inc"
