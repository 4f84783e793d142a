// mlp_h2f.hip -- the MLP of a ConvNeXt block with C = 128, hidden = 192 as ONE kernel (f16x2 arithmetic, pre-split operands):
//     out = x + gamma * ( fc2( gelu( fc1(y) + b1 ) ) + b2 )                      (lvae/models/common.py:131-132,154-158)
// for the decoder's eight stride-4 blocks (qarv/zoo.py:86-87: 8 x CNX(128, k7, mlp 1.5)).  As two launches (gemm_h2p.hip) these
// blocks are memory-side: fc1 writes and fc2 re-reads the 196608 x 192 hidden map (2 x 151 MB per block at batch 8) and run at
// 2.4 / 4.2 TB/s; they are the GPU tail of every decode (after the last latent block nothing else is left to overlap with).
// This is the one block shape whose weights are small enough to keep the hidden tile on the CU:
//   * a workgroup (8 waves) owns 128 rows.  Phase 1: its A rows (y, H2K32: 64 KB) and ALL of W1 (192 x 128, 96 KB) are fetched by
//     LDS-DMA -- 160 KB, the whole LDS -- and P = y W1^T is computed from LDS (wave tile 32 x 96);
//   * epilogue 1 (bias, exact-erf GELU, f16x2 split) writes the hidden tile back into LDS in the stage layout phase 2 reads its A
//     operand from (6 stages of 128 rows x 128 B, over the space A and W1 occupied), while W2's first four 32-deep stages arrive;
//   * phase 2: O = hidden W2^T (wave tile 32 x 64), W2 streaming through a 4-stage ring; epilogue 2: bias, gamma, residual, store.
// Arithmetic per element is the two-launch path's, operation for operation: per accumulator the MFMA sequence of gemm_h2p_kernel
// (k16 steps ascending; X: a_lo' w_hi, a_hi w_lo'; H: a_hi w_hi), fma(accX, 2^-11, accH), gemm_epilogue's "+ bias -> gelu" /
// "+ bias, * gamma, + residual" with the same roundings (no contraction), the same split_pair_h2 -- so every output bit equals
// fc2(fc1(.)) through gemm_h2p (tests/test_gpu_f16x2.py::test_mlp_h2f_equals_two_gemms), and the host may use it for this block
// shape at every batch size.
#include "gemm_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)
#define H2F_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

constexpr int F_C = 128, F_HID = 192, F_BM = 128;
constexpr int F_A_BYTES = 4 * F_BM * 128;             // A: 4 stages of 128 rows x 128 B
constexpr int F_W1_STAGE = F_HID * 128;               // 24 KB
constexpr int F_HID_BYTES = 6 * F_BM * 128;           // hidden: 6 stages of 128 rows x 128 B (96 KB)
constexpr int F_W2_STAGE = F_C * 128;                 // 16 KB, ring of 4 behind the hidden tile
constexpr int F_LDS = F_A_BYTES + 4 * F_W1_STAGE;     // 160 KB = F_HID_BYTES + 4 * F_W2_STAGE
static_assert(F_LDS == 160 * 1024 && F_HID_BYTES + 4 * F_W2_STAGE == F_LDS, "LDS map");

__global__ __launch_bounds__(512, 1) void mlp_h2f_kernel(const lvae_mlp_desc d) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * F_BM;
    const int rows = (d.M - m0) < F_BM ? (d.M - m0) : F_BM;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem);

    // per-column parameters of this lane's output columns (requested before the DMA queue fills: loads retire in order)
    float b1v[3], b2v[2], gmv[2];
#pragma unroll
    for (int b = 0; b < 3; ++b) b1v[b] = d.b1[96 * wn + 32 * b + li];
#pragma unroll
    for (int b = 0; b < 2; ++b) { b2v[b] = d.b2[64 * wn + 32 * b + li]; gmv[b] = d.gamma[64 * wn + 32 * b + li]; }

    // ---- DMA: instruction g of a stage covers rows 8g .. 8g + 7 (one 128-B line each); wave w issues g = i * 8 + w, so g has the
    // parity of w and the source permutation ((stage row >> 1) & 7 = (4 (w & 1) + (r_in >> 1)) & 7) is a per-lane constant
    const int r_in = lane >> 3, pp = lane & 7;
    const int perm = (pp ^ ((4 * (wave & 1) + (r_in >> 1)) & 7)) << 4;
    {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.y + (long)m0 * (F_C * 4)), 0, rows * (F_C * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)d.w1, 0, F_HID * F_C * 4, 0x00020000);
        const int dv = r_in * (F_C * 4) + perm;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int g = i * 8 + wave;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)((char*)smem + q * (F_BM * 128) + g * 1024), 16, dv,
                                                         8 * g * (F_C * 4) + q * 128, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int g = i * 8 + wave;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)((char*)smem + F_A_BYTES + q * F_W1_STAGE + g * 1024), 16,
                                                         dv, 8 * g * (F_C * 4) + q * 128, 0, 0);
            }
        }
    }
    // fragment addresses: piece (plane p, k16 step t, lane half) = 4p + 2t + lh at ((piece ^ x) << 4) of the lane's row
    const int xr = (li >> 1) & 7;
    unsigned po[4];                                                   // [2p + t]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) po[pt] = (unsigned)(((4 * (pt >> 1) + 2 * (pt & 1) + lh) ^ xr) << 4);
    // the hidden tile uses the piece permutation rot3((row >> 1) & 7) instead of (row >> 1) & 7: any bijection keeps the fragment reads
    // conflict-free, and this one also separates rows m and m + 2 in epilogue 1's ds_write_b64 pattern (4 rows x 4 column groups per
    // 16 lanes), which the plain form maps to the same banks (two-way conflicts on 31 % of the kernel's LDS cycles: r03_pmc_mlp_h2f.txt)
    const int xh = ((xr << 1) & 7) | (xr >> 2);
    unsigned ph[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) ph[pt] = (unsigned)(((4 * (pt >> 1) + 2 * (pt & 1) + lh) ^ xh) << 4);
    const unsigned a_row = lds0 + (32 * wm + li) * 128;               // A / hidden rows of this wave
    const unsigned w1_row = lds0 + F_A_BYTES + (96 * wn + li) * 128;
    const unsigned w2_row = lds0 + F_HID_BYTES + (64 * wn + li) * 128;

    // ---- phase 1: P[32 x 96] = y W1^T over K = 128; stage q is computed as soon as it has landed (five DMA instructions per wave and
    // stage, issued in stage order: "at most 5 (3 - q) outstanding" = my part of stage q is in LDS; the barrier makes it everyone's)
    f32x16 pH[3], pX[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { pH[b][r] = 0.f; pX[b][r] = 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q == 0) asm volatile("s_waitcnt vmcnt(15)\n\ts_barrier" ::: "memory");
        else if (q == 1) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
        else if (q == 2) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        LVAE_FENCE();
        f16x8 af[2][2], wf[2][3][2];                                  // [t][plane], [t][b][plane]
        const unsigned aq = a_row + q * (F_BM * 128), wq = w1_row + q * F_W1_STAGE;     // (the 16-bit offset field cannot hold these)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            H2F_DSR(af[t][0], aq + po[0 + t], 0);
            H2F_DSR(af[t][1], aq + po[2 + t], 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                H2F_DSR(wf[t][b][0], wq + po[0 + t], b * 4096);
                H2F_DSR(wf[t][b][1], wq + po[2 + t], b * 4096);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(wf[0][0][0]), "+v"(wf[0][0][1]),
                     "+v"(wf[0][1][0]), "+v"(wf[0][1][1]), "+v"(wf[0][2][0]), "+v"(wf[0][2][1]), "+v"(wf[1][0][0]), "+v"(wf[1][0][1]),
                     "+v"(wf[1][1][0]), "+v"(wf[1][1][1]), "+v"(wf[1][2][0]), "+v"(wf[1][2][1]));
        LVAE_FENCE();
#pragma unroll
        for (int t = 0; t < 2; ++t) {                                 // (column blocks inside: dependent MFMAs are three issues apart)
#pragma unroll
            for (int b = 0; b < 3; ++b) pX[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][1], wf[t][b][0], pX[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) pX[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][0], wf[t][b][1], pX[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) pH[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][0], wf[t][b][0], pH[b], 0, 0, 0);
        }
        LVAE_FENCE();
    }
    asm volatile("s_barrier" ::: "memory");                            // everyone is done reading A and W1
    LVAE_FENCE();

    // ---- W2: stages 0 .. 3 into the ring behind the hidden tile (they land while epilogue 1 runs)
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w2, 0, F_C * F_HID * 4, 0x00020000);
    const int dv2 = r_in * (F_HID * 4) + perm;
    auto dma_w2 = [&](int q) __attribute__((always_inline)) {          // stage q -> ring slot q & 3
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = i * 8 + wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (__attribute__((address_space(3))) void*)((char*)smem + F_HID_BYTES + (q & 3) * F_W2_STAGE + g * 1024),
                                                     16, dv2, 8 * g * (F_HID * 4) + q * 128, 0, 0);
        }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) dma_w2(q);

    // ---- epilogue 1: hidden = split(gelu(P + b1)) -> LDS, stage layout of an A operand (row m of stage c / 32: 64 B hi | 64 B lo',
    // 16-B pieces permuted by (m >> 1) & 7).  After the quad transpose a lane holds 4 consecutive columns of one row.
    {
        const int lj = li & 3;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int cs = 3 * wn + b;                                 // hidden columns 96 wn + 32 b .. + 31 = stage cs
            const int cc = li & ~3;                                    // first of the lane's 4 columns inside the stage
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = __builtin_fmaf(pX[b][4 * g + 0], 1.0f / 2048.0f, pH[b][4 * g + 0]) + b1v[b];
                float v1 = __builtin_fmaf(pX[b][4 * g + 1], 1.0f / 2048.0f, pH[b][4 * g + 1]) + b1v[b];
                float v2 = __builtin_fmaf(pX[b][4 * g + 2], 1.0f / 2048.0f, pH[b][4 * g + 2]) + b1v[b];
                float v3 = __builtin_fmaf(pX[b][4 * g + 3], 1.0f / 2048.0f, pH[b][4 * g + 3]) + b1v[b];
                gelu_erf2(v0, v1); gelu_erf2(v2, v3);
                quad_transpose(v0, v1, v2, v3, lj);
                unsigned h0, l0, h1, l1;
                split_pair_h2(v0, v1, h0, l0);
                split_pair_h2(v2, v3, h1, l1);
                const int m = 32 * wm + 4 * lh + 8 * g + lj;          // row (inside the tile) this lane now holds
                const int k = (m >> 1) & 7, x = ((k << 1) & 7) | (k >> 2);
                const unsigned base = lds0 + cs * (F_BM * 128) + m * 128 + ((cc & 7) << 1);
                const u32x2_t hi2 = {h0, h1}, lo2 = {l0, l1};
                asm volatile("ds_write_b64 %0, %1" ::"v"(base + ((((cc >> 3) + 0) ^ x) << 4)), "v"(hi2) : "memory");
                asm volatile("ds_write_b64 %0, %1" ::"v"(base + ((((cc >> 3) + 4) ^ x) << 4)), "v"(lo2) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // hidden tile complete, W2 stages 0 .. 3 landed
    LVAE_FENCE();

    // ---- phase 2: O[32 x 64] = hidden W2^T over K = 192
    f32x16 oH[2], oX[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oH[b][r] = 0.f; oX[b][r] = 0.f; }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        if (q == 2) {                                                  // ring slots 0 and 1 are free once everyone is past stages 0, 1
            asm volatile("s_barrier" ::: "memory");
            LVAE_FENCE();
            dma_w2(4); dma_w2(5);
        }
        if (q == 4) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            LVAE_FENCE();
        }
        f16x8 af[2][2], wf[2][2][2];
        const unsigned aq = a_row + q * (F_BM * 128), wq = w2_row + (q & 3) * F_W2_STAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            H2F_DSR(af[t][0], aq + ph[0 + t], 0);
            H2F_DSR(af[t][1], aq + ph[2 + t], 0);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                H2F_DSR(wf[t][b][0], wq + po[0 + t], b * 4096);
                H2F_DSR(wf[t][b][1], wq + po[2 + t], b * 4096);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(wf[0][0][0]), "+v"(wf[0][0][1]),
                     "+v"(wf[0][1][0]), "+v"(wf[0][1][1]), "+v"(wf[1][0][0]), "+v"(wf[1][0][1]), "+v"(wf[1][1][0]), "+v"(wf[1][1][1]));
        LVAE_FENCE();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int b = 0; b < 2; ++b) oX[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][1], wf[t][b][0], oX[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 2; ++b) oH[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][0], wf[t][b][0], oH[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 2; ++b) oX[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][0], wf[t][b][1], oX[b], 0, 0, 0);
        }
        LVAE_FENCE();
    }

    // ---- epilogue 2: out = res + gamma * (O + b2)   (gemm_epilogue's order: + bias, * gamma, transpose, + residual)
    {
        const int lj = li & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = m0 + 32 * wm + 4 * lh + 8 * g + lj;
            const bool rok = row < d.M;
            const long rbase = (long)(rok ? row : 0) * F_C;
            f32x4 rv[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) rv[b] = *(const f32x4*)(d.res + rbase + 64 * wn + 32 * b + (li & ~3));
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v0 = (__builtin_fmaf(oX[b][4 * g + 0], 1.0f / 2048.0f, oH[b][4 * g + 0]) + b2v[b]) * gmv[b];
                float v1 = (__builtin_fmaf(oX[b][4 * g + 1], 1.0f / 2048.0f, oH[b][4 * g + 1]) + b2v[b]) * gmv[b];
                float v2 = (__builtin_fmaf(oX[b][4 * g + 2], 1.0f / 2048.0f, oH[b][4 * g + 2]) + b2v[b]) * gmv[b];
                float v3 = (__builtin_fmaf(oX[b][4 * g + 3], 1.0f / 2048.0f, oH[b][4 * g + 3]) + b2v[b]) * gmv[b];
                quad_transpose(v0, v1, v2, v3, lj);
                if (rok) {
                    f32x4 o = {v0, v1, v2, v3};
                    o[0] += rv[b][0]; o[1] += rv[b][1]; o[2] += rv[b][2]; o[3] += rv[b][3];
                    *(f32x4*)(d.out + rbase + 64 * wn + 32 * b + (li & ~3)) = o;
                }
            }
        }
    }
}

}  // namespace

int lvae_mlp_h2c_try(const lvae_mlp_desc* d, hipStream_t st, int* rc);       // mlp_h2c.hip: the hidden-chunked form (weights streamed)

extern "C" int lvae_mlp_h2f(const lvae_mlp_desc* d, void* stream) {
    if (!d || !d->y || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->gamma || !d->res || !d->out || d->M <= 0) return -22;
    if (d->C != F_C || d->hid != F_HID) {
        int rc = 0;
        return lvae_mlp_h2c_try(d, (hipStream_t)stream, &rc) ? rc : -22;   // -22: no fused form of this block shape
    }
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)mlp_h2f_kernel, F_LDS)) return ae;
    hipLaunchKernelGGL(mlp_h2f_kernel, dim3((d->M + F_BM - 1) / F_BM), dim3(512), F_LDS, (hipStream_t)stream, *d);
    return (int)hipGetLastError();
}
