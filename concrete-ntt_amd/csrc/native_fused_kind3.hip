// whole-polymul kernel instantiations: native kind 3
#define INST_KIND 3
#include "native_fused_inst.inc"
