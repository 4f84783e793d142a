// The pre-split-output instances of dwconv_cl.hip (fp32 map in, f16x2 planes out: the A operand of csrc/gemm_h2p.hip) as a translation
// unit of their own, so that the instance sets compile in parallel.
#define LVAE_CL_H2_TU 1
#include "dwconv_cl.hip"
