// whole-polymul kernel instantiations: native kind 1
#define INST_KIND 1
#include "native_fused_inst.inc"
