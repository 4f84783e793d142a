// plan_runtime.cpp -- replay of a recorded launch plan segment from native code (lvae/engine.py: Plan.run).
//
// A codec's encode / decode for one (batch, height, width) is a flat list of launches with fully resolved arguments.  Python replays
// such a list with one ctypes call per launch (~10 us of interpreter time each, under the GIL); with several pipeline groups launching
// from their own threads -- and the rANS coder threads waking up in between -- the interpreter lock becomes the schedule.  Here a
// segment is ONE foreign call that holds no Python state: every entry names an entry point of this library (`kind`) and carries its
// arguments by class (pointers, integers, floats, each in call order); `side` entries go to the plan's side stream, ORDER entries record
// / wait the fork-join events between the two streams (lvae_stream_order).
#include <hip/hip_runtime.h>

#include <chrono>
#include <vector>

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
                                          (int)i[2], (int)i[3], (int*)p[4], st);
                break;
            case LVAE_OP_QUANTIZE:
                rc = lvae_quantize_f32((const float*)p[0], (const float*)p[1], (int32_t*)p[2], (float*)p[3], (int)i[0], (int)i[1], (int)i[2], (int)i[3],
                                       (int*)p[4], st);
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
                                              (int)i[0], (float)f[0], (int)i[1], (int)i[2], (int)i[3], (int*)p[6], st);
                break;
            case LVAE_OP_LOSSLESS_OUTPUT: rc = lvae_lossless_output_f32((const int32_t*)p[0], (const float*)p[1], (float*)p[2], i[0], (int*)p[3], st); break;
            case LVAE_OP_MLP_H2F: rc = lvae_mlp_h2f((const lvae_mlp_desc*)p[0], st); break;
            case LVAE_OP_MLP_SK: rc = lvae_mlp_sk((const lvae_mlp_sk_desc*)p[0], st); break;
            case LVAE_OP_PRIOR_INDEX_SK:
                rc = lvae_prior_index_sk_f32((const float*)p[0], (int)i[0], (const float*)p[1], (float*)p[2], (float*)p[3], (uint8_t*)p[4], (const float*)p[5],
                                             (int)i[1], (float)f[0], (int)i[2], (int)i[3], (int)i[4], (int*)p[6], st);
                break;
            case LVAE_OP_QUANTIZE_SK:
                rc = lvae_quantize_sk_f32((const float*)p[0], (int)i[0], (const float*)p[1], (float*)p[2], (const float*)p[3], (int32_t*)p[4], (float*)p[5],
                                          (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int*)p[6], st);
                break;
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


// rans_host.cpp: decode tables owned by the caller, so that the nine per-block coder calls of one decode build each table row once
struct LvaeDecTabs;
LvaeDecTabs* lvae_dec_tabs_new();
void lvae_dec_tabs_free(LvaeDecTabs* t);
int lvae_rans_decode_batch_tabs(int n_streams, const uint8_t* const* in, const size_t* in_len, const uint8_t* const* idx,
                                const size_t* n, const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                                int32_t* const* sym_out, int* status, int n_threads, LvaeDecTabs* tabs);

// rans_host.cpp: a block's streams handed to the coder pool without waiting for them
struct LvaeEncJob;
LvaeEncJob* lvae_rans_encode_batch_begin(int n_streams, const int32_t* const* sym, const uint8_t* const* idx, const size_t* n,
                                         const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                                         uint8_t* const* out, const size_t* out_cap, long* out_len, int n_threads);
int lvae_rans_encode_batch_end(LvaeEncJob* e);

namespace {
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// The decode loop of one pipeline group (lvae/models/*/model.py: decompress_batch) without the interpreter: the chain
// GPU segment -> indexes to the host -> rANS -> symbols to the device is latency-bound (nine dependent round trips per image), and with
// two groups decoding from two Python threads every step of it also queued for the interpreter lock.
static int decode_blocks_impl(const lvae_dec_block* blocks, int n_blocks, int n_images, const uint8_t* const* strings,
                              const size_t* string_len, const int32_t* qcdf, int row_stride, const int32_t* cdf_len,
                              const int32_t* offset, const lvae_op* tail_ops, int n_tail, const int* status_dev, int* status_host,
                              void* stream, void* side_stream, int n_threads, int* failed_block, int* failed_op, double* seconds) {
    if (!blocks || n_blocks < 0 || n_images <= 0 || !strings || !string_len || !qcdf || !cdf_len || !offset) return -22;
    hipStream_t st = (hipStream_t)stream;
    std::vector<const uint8_t*> idx_ptr(n_images);
    std::vector<int32_t*> out_ptr(n_images);
    std::vector<size_t> cnt(n_images);
    std::vector<int> status(n_images);
    struct TabsGuard { LvaeDecTabs* t; ~TabsGuard() { lvae_dec_tabs_free(t); } } tabs{lvae_dec_tabs_new()};
    if (!tabs.t) return -12;
    double t_gpu = 0.0, t_coder = 0.0;
    int bad = -1;
    // timeline request (header): seconds[0] = -(capacity in doubles, 8 ... 4096) AND seconds[1] = LVAE_TRACE_MAGIC on entry -> absolute
    // steady-clock stamps behind the two totals.  Both words are needed: `seconds` was output-only before the timeline existed, and a C
    // caller's uninitialised double[2] must never be taken for a capacity
    const int trace_cap = (seconds && seconds[0] <= -8.0 && seconds[0] >= -4096.0 && seconds[1] == LVAE_TRACE_MAGIC) ? (int)(-seconds[0]) : 0;
    auto stamp = [&](int slot, double t) { if (slot < trace_cap) seconds[slot] = t; };
    for (int b = 0; b < n_blocks; ++b) {
        const lvae_dec_block& k = blocks[b];
        const double t0 = now_s();
        int rc = lvae_run_ops(k.ops, k.n_ops, stream, side_stream, &bad);
        // (idx_dev == NULL: the segment's prior_index launch wrote the pinned host array itself -- include/lvae_hip.h, lvae_dec_block)
        if (rc == 0 && k.idx_dev) rc = (int)hipMemcpyAsync(k.idx_host, k.idx_dev, k.per_image * n_images, hipMemcpyDeviceToHost, st);
        const double t_issued = now_s();
        if (rc == 0) {
            // the segment is ~0.3 ms of GPU work and the coder is waiting for it: poll the stream for a bounded while (the wake-up of a
            // blocking wait is on the chain nine times per image: -0.05 ... 0.08 ms per decode, same-box), then block
            hipError_t q = hipErrorNotReady;
            for (int it = 0; it < 400000 && (q = hipStreamQuery(st)) == hipErrorNotReady; ++it) {}
            rc = q == hipErrorNotReady ? (int)hipStreamSynchronize(st) : (int)q;
        }
        if (rc != 0) {
            if (failed_block) *failed_block = b;
            if (failed_op) *failed_op = bad;
            return rc;
        }
        const double t1 = now_s();
        for (int i = 0; i < n_images; ++i) {
            idx_ptr[i] = k.idx_host + (size_t)i * k.per_image;
            out_ptr[i] = k.sym_host + (size_t)i * k.per_image;
            cnt[i] = k.per_image;
        }
        rc = lvae_rans_decode_batch_tabs(n_images, strings + (size_t)b * n_images, string_len + (size_t)b * n_images, idx_ptr.data(), cnt.data(),
                                         qcdf, row_stride, cdf_len, offset, out_ptr.data(), status.data(), n_threads, tabs.t);
        if (rc != 0) {
            // a stream that does not decode: corrupt / truncated -- or decoded against garbage scale indexes because a prior parameter was
            // non-finite (the indexes themselves are always valid table rows): the status word tells the two apart
            if (failed_block) *failed_block = b;
            // (on the group's own stream: a plain hipMemcpy goes through the legacy stream and would wait for the OTHER group's work too)
            if (status_dev && status_host && hipMemcpyAsync(status_host, status_dev, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
                hipStreamSynchronize(st) == hipSuccess && *status_host != 0)
                return -75;
            return -74;
        }
        // (sym_dev == NULL: the next segment's dequantize launch reads the pinned host array itself)
        if (k.sym_dev) rc = (int)hipMemcpyAsync(k.sym_dev, k.sym_host, k.per_image * n_images * sizeof(int32_t), hipMemcpyHostToDevice, st);
        if (rc != 0) {
            if (failed_block) *failed_block = b;
            return rc;
        }
        const double t2 = now_s();
        t_gpu += t1 - t0; t_coder += t2 - t1;
        stamp(2 + 4 * b, t0); stamp(3 + 4 * b, t_issued); stamp(4 + 4 * b, t1); stamp(5 + 4 * b, t2);
    }
    if (n_tail > 0) {
        int rc = lvae_run_ops(tail_ops, n_tail, stream, side_stream, &bad);
        // the status word travels behind the tail (no wait here: the caller reads *status_host after ITS synchronisation and must do so
        // before it hands the reconstruction on -- a NaN would otherwise pass the final clamp unseen)
        if (rc == 0 && status_dev && status_host) rc = (int)hipMemcpyAsync(status_host, status_dev, sizeof(int), hipMemcpyDeviceToHost, st);
        if (rc != 0) {
            if (failed_block) *failed_block = n_blocks;
            if (failed_op) *failed_op = bad;
            return rc;
        }
    }
    stamp(2 + 4 * n_blocks, now_s());
    if (seconds) { seconds[0] = t_gpu; seconds[1] = t_coder; }
    return 0;
}

static int encode_blocks_impl(const lvae_enc_block* blocks, int n_blocks, int n_images, uint8_t* const* out, const size_t* out_cap,
                              long* out_len, const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                              const int* status_dev, int* status_host, void* stream, void* side_stream, int n_threads,
                              int* failed_block, int* failed_op, double* seconds) {
    if (!blocks || n_blocks <= 0 || n_images <= 0 || !out || !out_cap || !out_len || !qcdf || !cdf_len || !offset) return -22;
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(n_blocks, nullptr);
    auto cleanup = [&]() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    const double t0 = now_s();
    int bad = -1;
    for (int b = 0; b < n_blocks; ++b) {
        const lvae_enc_block& k = blocks[b];
        int rc = lvae_run_ops(k.ops, k.n_ops, stream, side_stream, &bad);
        if (rc == 0 && k.sym_dev) rc = (int)hipMemcpyAsync(k.sym_host, k.sym_dev, k.per_image * n_images * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (rc == 0 && k.idx_dev) rc = (int)hipMemcpyAsync(k.idx_host, k.idx_dev, k.per_image * n_images, hipMemcpyDeviceToHost, st);
        // the status word travels ONCE, behind the last block's segment (a copy per block was measured: +0.1 ms per encode at batch 8)
        if (rc == 0 && b == n_blocks - 1 && status_dev && status_host) rc = (int)hipMemcpyAsync(status_host, status_dev, sizeof(int), hipMemcpyDeviceToHost, st);
        if (rc == 0) rc = (int)hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
        if (rc == 0) rc = (int)hipEventRecord(ev[b], st);
        if (rc != 0) {
            if (failed_block) *failed_block = b;
            if (failed_op) *failed_op = bad;
            (void)hipStreamSynchronize(st);
            cleanup();
            return rc;
        }
    }
    const double t1 = now_s();
    std::vector<const int32_t*> sym_ptr(n_images);
    std::vector<const uint8_t*> idx_ptr(n_images);
    std::vector<size_t> cnt(n_images);
    std::vector<LvaeEncJob*> jobs(n_blocks, nullptr);
    double t_wait = 0.0, t_coder = 0.0;
    int rc_all = 0, bad_block = -1;
    // A block's streams go to the coder pool the moment its event has fired, and this thread turns to the NEXT block's event at once:
    // the pool codes block b while the GPU computes block b + 1 and while block b - 1 may still be in the coder (the stride-16 blocks
    // take longer to code than the GPU takes for the block behind them) -- only what is still uncoded when the last event fires is
    // waited for (`seconds[2]`).
    for (int b = 0; b < n_blocks && rc_all == 0; ++b) {
        const lvae_enc_block& k = blocks[b];
        const double tw = now_s();
        int rc = (int)hipEventSynchronize(ev[b]);
        t_wait += now_s() - tw;
        // the status word behind the LAST block's segment: an out-of-range input (the reference's assert), or non-finite prior parameters /
        // posterior means (an fp16 overflow of the f16x2 arithmetic).  The earlier blocks are with the coder by now (it takes any int32
        // symbol and any scale index is a valid table row, so garbage cannot hurt it); the caller discards every string
        if (rc == 0 && b == n_blocks - 1 && status_dev && status_host && *status_host != 0) rc = (*status_host & LVAE_STATUS_RANGE) ? -34 : -75;
        if (rc == 0) {
            for (int i = 0; i < n_images; ++i) {
                sym_ptr[i] = k.sym_host + (size_t)i * k.per_image;
                idx_ptr[i] = k.idx_host + (size_t)i * k.per_image;
                cnt[i] = k.per_image;
            }
            jobs[b] = lvae_rans_encode_batch_begin(n_images, sym_ptr.data(), idx_ptr.data(), cnt.data(), qcdf, row_stride, cdf_len, offset,
                                                   out + (size_t)b * n_images, out_cap + (size_t)b * n_images, out_len + (size_t)b * n_images, n_threads);
            if (!jobs[b]) rc = -12;
        }
        if (rc != 0) { rc_all = rc; bad_block = b; }
    }
    const double tc = now_s();
    for (int b = 0; b < n_blocks; ++b) {                      // every job that was begun is ended (it owns heap state), error or not
        if (!jobs[b]) continue;
        const int rc = lvae_rans_encode_batch_end(jobs[b]);
        if (rc != 0 && rc_all == 0) { rc_all = rc; bad_block = b; }
    }
    t_coder = now_s() - tc;
    if (rc_all != 0) {
        if (failed_block) *failed_block = bad_block;
        (void)hipStreamSynchronize(st);
        cleanup();
        return rc_all;
    }
    cleanup();
    if (seconds) { seconds[0] = t1 - t0; seconds[1] = t_wait; seconds[2] = t_coder; }
    return 0;
}

// The C ABI never lets a C++ exception out (std::bad_alloc from the vectors / the coder's job objects would otherwise unwind through the
// foreign caller -- ctypes, cgo ... -- and end in std::terminate): -12 (ENOMEM) instead, after the group's stream has drained so that no
// launch of the failed call still reads the caller's buffers.
extern "C" int lvae_decode_blocks(const lvae_dec_block* blocks, int n_blocks, int n_images, const uint8_t* const* strings,
                                  const size_t* string_len, const int32_t* qcdf, int row_stride, const int32_t* cdf_len,
                                  const int32_t* offset, const lvae_op* tail_ops, int n_tail, const int* status_dev, int* status_host,
                                  void* stream, void* side_stream, int n_threads, int* failed_block, int* failed_op, double* seconds) {
    try {
        return decode_blocks_impl(blocks, n_blocks, n_images, strings, string_len, qcdf, row_stride, cdf_len, offset, tail_ops, n_tail, status_dev,
                                  status_host, stream, side_stream, n_threads, failed_block, failed_op, seconds);
    } catch (...) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        return -12;
    }
}

extern "C" int lvae_encode_blocks(const lvae_enc_block* blocks, int n_blocks, int n_images, uint8_t* const* out, const size_t* out_cap,
                                  long* out_len, const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                                  const int* status_dev, int* status_host, void* stream, void* side_stream, int n_threads,
                                  int* failed_block, int* failed_op, double* seconds) {
    try {
        return encode_blocks_impl(blocks, n_blocks, n_images, out, out_cap, out_len, qcdf, row_stride, cdf_len, offset, status_dev, status_host,
                                  stream, side_stream, n_threads, failed_block, failed_op, seconds);
    } catch (...) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        return -12;
    }
}
