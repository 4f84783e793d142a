// The bf16-map instances of dwconv_cl.hip (reduced-precision mode, BASELINE config 5) as a translation unit of their own, so that the
// two halves of the instance set compile in parallel.
#define LVAE_CL_BF16_TU 1
#include "dwconv_cl.hip"
