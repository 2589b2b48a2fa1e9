// whole-polymul kernel instantiations: native kind 0
#define INST_KIND 0
#include "native_fused_inst.inc"
