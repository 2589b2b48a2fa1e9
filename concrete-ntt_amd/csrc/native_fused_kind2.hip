// whole-polymul kernel instantiations: native kind 2 (native128::Plan32)
#define INST_KIND 2
#include "native_fused_inst.inc"
