// plan_runtime.cpp -- replay of a recorded launch plan segment from native code (lvae/engine.py: Plan.run).
//
// A codec's encode / decode for one (batch, height, width) is a flat list of launches with fully resolved arguments.  Python replays
// such a list with one ctypes call per launch (~10 us of interpreter time each, under the GIL); with several pipeline groups launching
// from their own threads -- and the rANS coder threads waking up in between -- the interpreter lock becomes the schedule.  Here a
// segment is ONE foreign call that holds no Python state: every entry names an entry point of this library (`kind`) and carries its
// arguments by class (pointers, integers, floats, each in call order); `side` entries go to the plan's side stream, ORDER entries record
// / wait the fork-join events between the two streams (lvae_stream_order).
#include <hip/hip_runtime.h>

#include "../../include/lvae_hip.h"

extern "C" int lvae_run_ops(const lvae_op* ops, int n, void* stream, void* side_stream, int* failed_index) {
    if (!ops || n < 0) return -22;
    for (int k = 0; k < n; ++k) {
        const lvae_op& o = ops[k];
        void* st = o.side ? side_stream : stream;
        void* const* p = o.p;
        const long* i = o.i;
        const double* f = o.f;
        int rc = -22;
        switch (o.kind) {
            case LVAE_OP_GEMM: rc = lvae_gemm_f32((const lvae_gemm_desc*)p[0], st); break;
            case LVAE_OP_DWCONV_LN_F32:
                rc = lvae_dwconv_ln_f32((const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                        (const float*)p[5], (const float*)p[6], (float*)p[7], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], st);
                break;
            case LVAE_OP_DWCONV_LN_H2:
                rc = lvae_dwconv_ln_h2((const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                       (const float*)p[5], (const float*)p[6], p[7], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], st);
                break;
            case LVAE_OP_DWCONV_LN_BF16:
                rc = lvae_dwconv_ln_bf16(p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4], (const float*)p[5],
                                         (const float*)p[6], p[7], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], st);
                break;
            case LVAE_OP_DWCONV_LN_Q8:
                rc = lvae_dwconv_ln_q8(p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4], (const float*)p[5],
                                       (const float*)p[6], p[7], (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], st);
                break;
            case LVAE_OP_STEM_F32:
                rc = lvae_stem_f32((const float*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[3], (int)i[0], (int)i[1], (int)i[2],
                                   (int)i[3], (float)f[0], (float)f[1], (int*)p[4], st);
                break;
            case LVAE_OP_STEM_BF16:
                rc = lvae_stem_bf16((const float*)p[0], (const float*)p[1], (const float*)p[2], p[3], (int)i[0], (int)i[1], (int)i[2], (int)i[3],
                                    (float)f[0], (float)f[1], (int*)p[4], st);
                break;
            case LVAE_OP_BIAS_EXPAND_F32: rc = lvae_bias_expand_f32((const float*)p[0], (float*)p[1], i[0], (int)i[1], st); break;
            case LVAE_OP_BIAS_EXPAND_BF16: rc = lvae_bias_expand_bf16((const float*)p[0], p[1], i[0], (int)i[1], st); break;
            case LVAE_OP_PRIOR_INDEX:
                rc = lvae_prior_index_f32((const float*)p[0], (float*)p[1], (uint8_t*)p[2], (const float*)p[3], (int)i[0], (float)f[0], (int)i[1],
                                          (int)i[2], (int)i[3], st);
                break;
            case LVAE_OP_QUANTIZE:
                rc = lvae_quantize_f32((const float*)p[0], (const float*)p[1], (int32_t*)p[2], (float*)p[3], (int)i[0], (int)i[1], (int)i[2], (int)i[3], st);
                break;
            case LVAE_OP_DEQUANTIZE:
                rc = lvae_dequantize_f32((const int32_t*)p[0], (const float*)p[1], (float*)p[2], (int)i[0], (int)i[1], (int)i[2], (int)i[3], st);
                break;
            case LVAE_OP_GAUSSIAN_NLL:
                rc = lvae_gaussian_nll_f32((const float*)p[0], (const int32_t*)p[1], (double*)p[2], (float)f[0], (int)i[0], (int)i[1], (int)i[2],
                                           (int)i[3], st);
                break;
            case LVAE_OP_LOSSLESS_PARAMS:
                rc = lvae_lossless_params_f32((const float*)p[0], (const float*)p[1], (float*)p[2], (uint8_t*)p[3], (int32_t*)p[4], (const float*)p[5],
                                              (int)i[0], (float)f[0], (int)i[1], (int)i[2], (int)i[3], st);
                break;
            case LVAE_OP_LOSSLESS_OUTPUT: rc = lvae_lossless_output_f32((const int32_t*)p[0], (const float*)p[1], (float*)p[2], i[0], st); break;
            case LVAE_OP_ORDER:      // i[0] != 0: the side stream waits for the main stream (fork); else the main stream for the side stream (join)
                rc = i[0] ? lvae_stream_order(stream, side_stream, p[0]) : lvae_stream_order(side_stream, stream, p[0]);
                break;
            default: rc = -22;
        }
        if (rc != 0) {
            if (failed_index) *failed_index = k;
            return rc;
        }
    }
    return 0;
}
