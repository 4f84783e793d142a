// The 2x2-patch gather instances of gemm_f32.hip as a translation unit of their own (compile time).
#define LVAE_GEMM_TU_AMODE 1
#include "gemm_f32.hip"
