// The 3x3-tap gather instances of gemm_f32.hip as a translation unit of their own (compile time).
#define LVAE_GEMM_TU_AMODE 2
#include "gemm_f32.hip"
