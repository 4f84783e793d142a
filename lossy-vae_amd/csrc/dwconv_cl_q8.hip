// The MX-fp8-output instances of dwconv_cl.hip (bf16 map in, e4m3 + E8M0 block scales out: the A operand of csrc/gemm_q8.hip in the
// reduced-precision mode) as a translation unit of their own, so that the instance sets compile in parallel.
#define LVAE_CL_Q8_TU 1
#include "dwconv_cl.hip"
