// whole-polymul kernel instantiations: native kind 4
#define INST_KIND 4
#include "native_fused_inst.inc"
