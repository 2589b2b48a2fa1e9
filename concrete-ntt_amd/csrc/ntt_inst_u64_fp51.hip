// CLS_FP51 kernel instantiations (64-bit words, 2^50 <= p < 2^51): as ntt_inst_u64_fp.hip with the denser range reductions.
#define INST_FPCLS CLS_FP51
#include "ntt_fp_inst.inc"
