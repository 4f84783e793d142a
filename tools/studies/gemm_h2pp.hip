// gemm_h2pp.hip -- STUDY, not part of liblvae_hip.so (tools/build_exp.sh with EXTRA_SRC=gemm_h2pp.hip -DLVAE_EXP_H2PP; tile codes 92 / 91):
// persistent form of gemm_h2p.hip's f16x2 GEMM on pre-split operands (prec 4, a_h2 = 1; H2K32 planes by LDS-DMA).  Bit-identical to
// gemm_h2p_kernel, and slower: see "Result" below and DESIGN.md 5c.
//
// Why.  gemm_h2p_kernel runs two workgroups per CU so that one's epilogue can overlap the other's main loop -- but the matrix pipe and
// the vector ALU share each SIMD's issue port ACROSS waves: while one wave's MFMA executes, only instructions of the SAME wave issue in
// its shadow (measured: ubench/mfma_valu_interleave), so two co-resident workgroups serialise epilogue and main loop instead
// of overlapping them, and with K = 192 ... 768 (the MLP of the large maps) the epilogue (bias, exact-erf GELU, f16x2 split, stores) is
// 30-45 % of a launch.  Here ONE workgroup of four waves owns a CU (one wave per SIMD) and walks a static list of tiles:
//   * the LDS-DMA pipeline is flat across tiles (4 stages of 32 k, 128 KB): the first stages of the next tile are in flight while the
//     current one finishes, so no tile pays a pipeline fill;
//   * fragments of k16 step 0 of stage f + 1 are read during step 1 of stage f (both stages have landed at stage f's barrier), so no MFMA
//     waits for LDS;
//   * OVERLAP: two accumulator sets; the epilogue of tile j - 1 (set P) is cut into 16 units (32 rows x 4 row groups x 2 column blocks)
//     that are issued between the MFMAs of the first stages of tile j (set S) -- same wave, so they do fill the MFMA shadows.
// Result (MI355X, profiles/r03_gemm_h2pp_study.txt).  Stage 1 (flat pipeline + cross-barrier fragment prefetch, epilogue NOT yet
// overlapped) runs the K = 4096 main loop at 315-320 TFLOP/s against 367-379 for two co-resident 128 x 128 workgroups, and the ablations
// say why: MFMAs + barriers alone 645-691 (the matrix-pipe rate at the ~2.0 GHz this load clocks at), + the 16 ds_read_b128 per stage
// 451-464, + the 8 LDS-DMA instructions per stage instead 346-382, both 320; dropping the vmcnt wait or the barrier changes nothing.  With
// one wave per SIMD every LDS-DMA issue (~75 cycles) and ds_read (~25) is exposed -- a second wave's MFMAs are what hides them -- so the
// five-odd issue slots per MFMA the epilogue units were to fill are already oversubscribed by the loop's own traffic.  The OVERLAP half was
// therefore not built; the file stays as the record of the measurement.
// Arithmetic per output element is gemm_h2p_kernel's (same MFMA sequence per accumulator, same epilogue expression), so every output
// bit equals gemm_h2p_kernel's and gemm_h2_kernel's.
#include "../../lossy-vae_amd/csrc/gemm_common.h"

#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef LVAE_EXP_PP_NODSR          // timing studies (wrong results): what is left without fragment reads / DMA / DMA waits / barriers
#define H2PP_DSR(dst, addr, off) asm volatile("; no ds_read %0 %1 %2" : "=v"(dst) : "v"(addr), "n"(off))
#else
#define H2PP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif
#if defined(LVAE_EXP_PP_NOBAR)
#define H2PP_STAGE_SYNC(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#elif defined(LVAE_EXP_PP_NOWAIT)
#define H2PP_STAGE_SYNC(n) asm volatile("s_barrier" ::: "memory")
#else
#define H2PP_STAGE_SYNC(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(n) : "memory")
#endif

template <int TN>
__global__ __launch_bounds__(256, 1) void gemm_h2pp_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    using C = Cfg<2, 2, 2, TN, 1, 32>;
    constexpr int BM = 128, BN = 64 * TN, ROWS = BM + BN, STAGE = ROWS * 128, NBUF = 4;
    constexpr int NWAVE = 4, NG = ROWS / 8, NI = NG / NWAVE, NIA = BM / 8 / NWAVE;       // DMA instructions per stage; per wave; of them A rows
    static_assert(NG % NWAVE == 0 && NI <= 12, "whole DMA instructions per wave, at most one per MFMA pair");
    static_assert(NBUF * STAGE <= 160 * 1024, "LDS");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int G = gridDim.x, bid = blockIdx.x;
    const int q8 = n_tiles / 8, r8 = n_tiles % 8;
    // tile list of this workgroup: indices bid, bid + G, ... ; index -> tile keeps the tiles of one XCD contiguous (G % 8 == 0 or a single
    // tile per workgroup, so every tile of a workgroup is in its XCD's chunk)
    auto tile_of = [&](int idx) __attribute__((always_inline)) {
        const int xcd = idx % 8, loc = idx / 8;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    };
    const int ntl = (n_tiles - bid + G - 1) / G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nq = d.K / 32;
    const int rowb = d.K * 4;                                       // bytes of one H2K32 row (A and W alike)
    const int dM = d.M, dN = d.N;
    const char* const dA = (const char*)d.A0;
    const char* const dW = (const char*)d.Wt16;

    // ---- DMA side: wave w issues instructions g = i * NWAVE + w of a stage (rows 8g .. 8g + 7; A rows first), i < NIA from A
    const int r_in = lane >> 3, pp = lane & 7;
    const int dvoff = r_in * rowb + ((pp ^ ((4 * (wave & 1) + (r_in >> 1)) & 7)) << 4);
    int jd = 0, sd = 0;                                              // tile (list position) and stage the next DMA fetches
    int m0d, n0d;
    {
        const int t = tile_of(bid);
        const int tm = t / tiles_n;
        m0d = tm * BM; n0d = (t - tm * tiles_n) * BN;
    }
    auto dma = [&](int i, int buf) __attribute__((always_inline)) {   // i: compile-time after unrolling; buf: uniform
        const int g = i * NWAVE + wave;
#ifdef LVAE_EXP_PP_NODMA
        return;
#endif
        if (i < NIA) {
            const int rows_a = (dM - m0d) < BM ? (dM - m0d) : BM;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dA + (long)m0d * rowb), 0, rows_a * rowb, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((char*)smem + buf * STAGE + g * 1024), 16, dvoff,
                                                     8 * g * rowb + sd * 128, 0, 0);
        } else {
            const int rows_w = (dN - n0d) < BN ? (dN - n0d) : BN;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dW + (long)n0d * rowb), 0, rows_w * rowb, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((char*)smem + buf * STAGE + g * 1024), 16, dvoff,
                                                     (8 * g - BM) * rowb + sd * 128, 0, 0);
        }
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {        // beyond the last stage: keep re-reading it (uniform vmcnt arithmetic)
        if (sd + 1 < nq) { ++sd; return; }
        if (jd + 1 < ntl) {
            ++jd; sd = 0;
            const int t = tile_of(bid + jd * G);
            const int tm = t / tiles_n;
            m0d = tm * BM; n0d = (t - tm * tiles_n) * BN;
        }
    };

    // ---- fragment side.  Piece (plane p, k16 step t, lane half lh) = 4p + 2t + lh, read at ((piece ^ x) << 4) of the lane's row.
    const int xr = (li >> 1) & 7;
    unsigned a_base[4], b_base[4];                                  // [2p + t]: byte address inside a stage
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int piece = 4 * (pt >> 1) + 2 * (pt & 1) + lh;
        const unsigned o = (unsigned)((piece ^ xr) << 4);
        a_base[pt] = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem) + (wave_m * 64 + li) * 128 + o;
        b_base[pt] = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem) + (BM + wave_n * 32 * TN + li) * 128 + o;
    }

    f32x16 accH[2][TN], accX[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accH[a][b][r] = 0.f; accX[a][b][r] = 0.f; }

    f16x8 f0a[2][2][2], f0b[2][TN][2];                              // k16 step 0 fragments, two sets: [set][a | b][plane]
    f16x8 f1a[2][2], f1b[TN][2];                                    // k16 step 1 fragments
    auto read_t0 = [&](auto set_tag, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        (void)f0a; (void)f0b;          // (operands of asm statements alone do not make a generic lambda capture a variable)
        const unsigned o = (unsigned)(buf * STAGE);
        const unsigned a0 = a_base[0] + o, a2 = a_base[2] + o, b0 = b_base[0] + o, b2 = b_base[2] + o;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            H2PP_DSR(f0a[S][a][0], a0, a * 4096);
            H2PP_DSR(f0a[S][a][1], a2, a * 4096);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            H2PP_DSR(f0b[S][b][0], b0, b * 4096);
            H2PP_DSR(f0b[S][b][1], b2, b * 4096);
        }
    };
    auto wait_t0 = [&](auto set_tag, auto n_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value, N = decltype(n_tag)::value;
        (void)f0a; (void)f0b;
        if constexpr (TN == 2)
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f0a[S][0][0]), "+v"(f0a[S][0][1]), "+v"(f0a[S][1][0]), "+v"(f0a[S][1][1]),
                         "+v"(f0b[S][0][0]), "+v"(f0b[S][0][1]), "+v"(f0b[S][1][0]), "+v"(f0b[S][1][1]) : "n"(N));
        else
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f0a[S][0][0]), "+v"(f0a[S][0][1]), "+v"(f0a[S][1][0]), "+v"(f0a[S][1][1]),
                         "+v"(f0b[S][0][0]), "+v"(f0b[S][0][1]) : "n"(N));
    };

    // prologue: stages 0, 1, 2 in flight; stage 0 landed for everyone; its step-0 fragments requested
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) {
#pragma unroll
        for (int i = 0; i < NI; ++i) dma(i, s);
        dma_advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NI) : "memory");
    LVAE_FENCE();
    read_t0(std::integral_constant<int, 0>{}, 0);

    // One 32-deep stage; f = flat stage index, f & 3 = its LDS buffer; its step-0 fragments are (being) read into set FS = f & 1.
    auto stage_body = [&](auto fs_tag, int f, bool last_of_tile) __attribute__((always_inline)) {
        constexpr int FS = decltype(fs_tag)::value;
        const int buf = f & 3, NXT = (f + 3) & 3, BUF1 = (f + 1) & 3;
        // my DMA instructions of the NEXT stage have landed once at most one later stage's are outstanding; after the barrier everyone's
        // have, and everyone is done reading the previous stage's buffer (= the one the stage three ahead goes to)
        H2PP_STAGE_SYNC(NI);
        LVAE_FENCE();
        {
            const unsigned o = (unsigned)(buf * STAGE);
            const unsigned a1 = a_base[1] + o, a3 = a_base[3] + o, b1 = b_base[1] + o, b3 = b_base[3] + o;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                H2PP_DSR(f1a[a][0], a1, a * 4096);
                H2PP_DSR(f1a[a][1], a3, a * 4096);
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                H2PP_DSR(f1b[b][0], b1, b * 4096);
                H2PP_DSR(f1b[b][1], b3, b * 4096);
            }
        }
        wait_t0(std::integral_constant<int, FS>{}, std::integral_constant<int, 4 + 2 * TN>{});
        LVAE_FENCE();
        int issued = 0;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j == 0) {
                    accX[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0a[FS][0][1], f0b[FS][b][0], accX[0][b], 0, 0, 0);
                    accX[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0a[FS][1][1], f0b[FS][b][0], accX[1][b], 0, 0, 0);
                } else if (j == 1) {
                    accX[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0a[FS][0][0], f0b[FS][b][1], accX[0][b], 0, 0, 0);
                    accX[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0a[FS][1][0], f0b[FS][b][1], accX[1][b], 0, 0, 0);
                } else {
                    accH[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0a[FS][0][0], f0b[FS][b][0], accH[0][b], 0, 0, 0);
                    accH[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0a[FS][1][0], f0b[FS][b][0], accH[1][b], 0, 0, 0);
                }
                if (issued < NI) { dma(issued, NXT); ++issued; }
                LVAE_FENCE();
            }
        }
        if constexpr (TN == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f1a[0][0]), "+v"(f1a[0][1]), "+v"(f1a[1][0]), "+v"(f1a[1][1]),
                         "+v"(f1b[0][0]), "+v"(f1b[0][1]), "+v"(f1b[1][0]), "+v"(f1b[1][1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f1a[0][0]), "+v"(f1a[0][1]), "+v"(f1a[1][0]), "+v"(f1a[1][1]), "+v"(f1b[0][0]), "+v"(f1b[0][1]));
        LVAE_FENCE();
        read_t0(std::integral_constant<int, FS ^ 1>{}, BUF1);      // the next stage has landed (barrier above)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j == 0) {
                    accX[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1a[0][1], f1b[b][0], accX[0][b], 0, 0, 0);
                    accX[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1a[1][1], f1b[b][0], accX[1][b], 0, 0, 0);
                } else if (j == 1) {
                    accX[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1a[0][0], f1b[b][1], accX[0][b], 0, 0, 0);
                    accX[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1a[1][0], f1b[b][1], accX[1][b], 0, 0, 0);
                } else {
                    accH[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1a[0][0], f1b[b][0], accH[0][b], 0, 0, 0);
                    accH[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1a[1][0], f1b[b][0], accH[1][b], 0, 0, 0);
                }
                if (issued < NI) { dma(issued, NXT); ++issued; }
                LVAE_FENCE();
            }
        }
        dma_advance();
        // the compiler must not see live, still-landing fragment registers across the epilogue's code: settle them first
        if (last_of_tile) { wait_t0(std::integral_constant<int, FS ^ 1>{}, std::integral_constant<int, 0>{}); LVAE_FENCE(); }
    };

    int f = 0;
    for (int j = 0; j < ntl; ++j) {
        const int t = tile_of(bid + j * G);
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
        for (int s = 0; s < nq; s += 2, f += 2) {                    // nq is even (K % 64 == 0): a tile starts with fragment set 0
            stage_body(std::integral_constant<int, 0>{}, f, false);
            stage_body(std::integral_constant<int, 1>{}, f + 1, s + 2 == nq);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) accH[a][b][r] = __builtin_fmaf(accX[a][b][r], 1.0f / 2048.0f, accH[a][b][r]);
        gemm_finish<C>(d, accH, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { accH[a][b][r] = 0.f; accX[a][b][r] = 0.f; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // trailing (redundant) DMAs have landed before LDS is freed
}

template <int TN>
int launch_h2pp(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int BM = 128, BN = 64 * TN, LDS = 4 * (BM + BN) * 128;
    static bool attr_set = false;
    static int ncu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_h2pp_kernel<TN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return (int)e;
        if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
        ncu -= ncu % 8;                                               // whole XCD rounds: a workgroup's tiles stay in its XCD's chunk
        if (ncu <= 0) return -22;
        attr_set = true;
    }
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_h2pp_kernel<TN>), dim3(n_tiles < ncu ? n_tiles : ncu), dim3(256), LDS, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// Entry point for gemm_h2p.hip's chooser.  tn: 2 = 128 x 128 tiles, 1 = 128 x 64.  Preconditions are lvae_gemm_h2p_try's, plus K >= 128, K % 64 == 0.
int lvae_gemm_h2pp_launch(const lvae_gemm_desc* d, hipStream_t st, int tn) {
    if (d->K < 128 || (d->K & 63)) return -22;
    return tn == 2 ? launch_h2pp<2>(d, st) : launch_h2pp<1>(d, st);
}
