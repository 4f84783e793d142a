// gemm_h2p.hip -- f16x2 GEMM (prec 4) whose BOTH operands arrive pre-split in the plane format H2K32 = [rows][K/32][2][32] fp16
// (include/lvae_hip.h: lvae_gemm_desc.a_h2): the MLP of every ConvNeXt block, fc1(y) and fc2(gelu(fc1)) (lvae/models/common.py:
// 131-132,154), where the A operand has exactly one consumer and its producer (the depthwise+LayerNorm kernel, fc1's GELU epilogue)
// can store hi / lo' planes instead of fp32 at the same 4 bytes per element.
//
// Why.  gemm_h2_kernel (gemm_h2.hip) halves the matrix-pipe time of the bf16x3 arithmetic but still converts A inside its main loop:
// per 16-deep stage a wave issues 12 MFMAs against ~50 VALU / LDS / VMEM instructions (fp32 -> hi, lo' split, ds_writes, register
// staging), and with two waves per SIMD every one of them competes with the other wave's MFMAs for the issue port: 220-240 TFLOP/s on
// the large layers, 27 % of the 3-MFMA roof (profiles/r03_*).  With both operands already in MFMA operand form the main loop is
// LDS-DMA + fragment reads + MFMAs only:
//   * global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write, no conversion): one wave instruction moves
//     8 rows x 128 B -- whole cache lines (a 32-deep stage of a row IS one 128-B line in H2K32); fragment-shaped 64-B pieces keep the
//     texture-address unit twice as busy for the same bytes (cdna_hip_programming.md, "x operand through LDS in full 128-B lines");
//   * the LDS image is lane-linear (DMA writes base + 16 * lane), so bank conflicts are avoided by permuting the SOURCE address: the
//     16-B piece at physical position pp of stage row r holds logical piece pp ^ ((r >> 1) & 7); a ds_read_b128 lane group
//     ({0-3,12-15,20-27}, ... : MI355X_MICROARCH.md) then covers 16 distinct bank quads for every fragment read;
//   * NBUF LDS stages of 32 k (3 x 48 KB for the 256 x 128 tile), DMA NBUF - 1 stages ahead, ONE raw s_barrier per stage with a counted
//     vmcnt (a __syncthreads would drain the DMA queue);
//   * fragment reads are inline asm (hipcc would guard every LDS load that may alias a DMA in flight with vmcnt(0)) with exact lgkmcnt
//     waits; the DMA instructions of the stage two ahead are spread between the MFMAs.
// Tiles: 64*WM x 64*TN, 2*WM waves (wave tile 64 x 32*TN, two accumulator sets): WM = 4 -> 256 rows, one workgroup per CU;
// WM = 2 -> 128 rows, two per CU.  Per accumulator the MFMA sequence is gemm_h2_kernel's (k16 steps ascending; X: a_lo'*w_hi, a_hi*w_lo';
// H: a_hi*w_hi), and the producers' split is the consumer's (exact), so every output bit equals gemm_h2_kernel's on the fp32 operand.
#include "gemm_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)
// timing ablations (wrong results by construction; tools/build_exp.sh only): what a launch costs without its MFMAs / fragment reads / epilogue
#if !defined(LVAE_EXPERIMENTAL_BUILD) && (defined(H2P_EXP_NOMFMA) || defined(H2P_EXP_NODSR) || defined(H2P_EXP_NOEPI) || defined(H2P_EXP_NODMA))
#error "H2P_EXP_* ablations need -DLVAE_EXPERIMENTAL_BUILD (tools/build_exp.sh)"
#endif
#ifdef H2P_EXP_NODSR
#define H2P_DSR(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr))
#else
#define H2P_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif
#ifdef H2P_EXP_NOMFMA
#define H2P_MFMA(a, b, c) (c)
#else
#define H2P_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

// (the straight-line epilogue h2p_epilogue_fast lives in gemm_common.h)
// FOLD ("serial split-K", d.ksplit = S > 1 with a_h2): ONE workgroup walks the S contiguous K slices of its tile and adds their partial sums
// in slice order -- tot = P_0; tot += P_1; ... with P_s = accH_s + accX_s / 2048 -- then applies splitk_epilogue_store: the operations
// of the parallel form (gemm_h2_kernel with gridDim.y = S + the reduce kernel / last arriver) in the same order, hence the same bits,
// without workspace traffic or a second launch.  The host takes it when the BATCH gives enough tiles to fill the chip; the slice count
// itself stays a function of the per-image shape (lvae/engine.py::auto_ksplit), so batched and single-image calls agree bit for bit.
// NLOAD > 0 (round 6): NLOAD extra LOADER waves issue every LDS-DMA piece and the 2 WM compute waves none.  For the few-tile, long-K
// launches (serial split-K on the stride-32 / 64 maps: <= one workgroup per CU, 48-64 stages) the stage time WAS the compute waves' DMA
// issue: an LDS-DMA piece costs its wave ~180 cycles (profiles/r06_cu_ingest.txt), six pieces per wave and stage = 1080 cycles against
// 384 cycles of MFMAs, with no second workgroup on the CU to hide them.  Eight loaders issue three pieces each; the compute waves' stage
// is barrier -> fragment reads -> MFMAs.  One s_barrier per stage joins all waves: the loaders arrive when the stage has landed (counted
// vmcnt), the compute waves when they are done with the stage before, and behind it the loaders refill that stage's slot.  Same MFMA
// sequence per accumulator, same fold, same epilogue: same bits (tests/test_gpu_f16x2.py::test_gemm_h2p_serial_split_k_*).
template <int WM, int TN, int NBUF, bool FOLD = false, int NLOAD = 0>
__global__ __launch_bounds__(128 * WM + 64 * NLOAD, (NLOAD ? 1 : (WM == 4 ? 1 : (NBUF == 2 && TN == 1 ? 3 : 2)))) void gemm_h2p_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles, int stagger) {
    using C = Cfg<WM, 2, 2, TN, 1, 32>;
    constexpr int BM = 64 * WM, BN = 64 * TN, ROWS = BM + BN, STAGE = ROWS * 128;
    constexpr int NWAVE = 2 * WM, NG = ROWS / 8, NI = NG / NWAVE;       // DMA instructions per stage, per wave and stage
    static_assert(NG % NWAVE == 0 && NWAVE % 2 == 0, "whole DMA instructions per wave; g has the parity of the wave");
    static_assert(NBUF * STAGE <= 160 * 1024 / (NLOAD ? 1 : (WM == 4 ? 1 : (NBUF == 2 && TN == 1 ? 3 : 2))), "LDS");
    static_assert(NLOAD == 0 || (NLOAD % 2 == 0 && NG % NLOAD == 0 && (NBUF - 2) * (NG / (NLOAD ? NLOAD : 1)) <= 63), "whole pieces per loader wave; the piece index has the loader's parity");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nq = d.K / 32;
    const int rowb = d.K * 4;                                       // bytes of one H2K32 row (A and W alike)
    // Phase stagger (LVAE_H2P_STAGGER, off by default).  Workgroups of one launch start together and take equally long, so the chip
    // alternates between main loops (matrix pipe busy, HBM nearly idle) and epilogues (every CU storing its tile, the matrix pipe idle).
    // Starting every second workgroup half a main loop late -- the second resident workgroup of a CU (told by its LDS allocation
    // base) or, with one workgroup per CU, the odd CUs -- was measured and did NOT help (0 ... -5 %): kept as a knob for the record.
    if (stagger > 0) {
        const bool late = WM == 4 ? ((__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (8 << 6) | 4) & 1) != 0)          // HW_ID.CU_ID bit 0
                                  : (__builtin_amdgcn_s_getreg(((8 - 1) << 11) | (0 << 6) | 6) != 0);                // LDS_ALLOC.LDS_BASE
        if (late)
            for (int i = 0; i < nq * stagger; ++i) __builtin_amdgcn_s_sleep(8);        // 512 cycles each
    }

    // ---- DMA side.  Wave w issues the stage's instructions g = i * NWAVE + w (i < NI): rows 8g .. 8g + 7 of the stage (A rows first).
    const int rows_a = (d.M - m0) < BM ? (d.M - m0) : BM, rows_w = (d.N - n0) < BN ? (d.N - n0) : BN;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.A0 + (long)m0 * rowb), 0, rows_a * rowb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.Wt16 + (long)n0 * rowb), 0, rows_w * rowb, 0x00020000);
    if constexpr (NLOAD > 0) {
        if (wave >= NWAVE) {
            // ---- loader waves: piece g = i * NLOAD + lw of every stage (rows 8 g .. 8 g + 7: A rows first), NBUF - 1 stages ahead
            const int lw = wave - NWAVE;
            constexpr int PL = NG / NLOAD;
            const int lr = lane >> 3, lp = lane & 7;
            const int lvoff = lr * rowb + ((lp ^ ((4 * (lw & 1) + (lr >> 1)) & 7)) << 4);
            auto issue = [&](int stage, int buf) {
#pragma unroll
                for (int i = 0; i < PL; ++i) {
                    const int g = i * NLOAD + lw;
                    const bool isA = g < BM / 8;                    // uniform
                    const int soff = (isA ? 8 * g : 8 * g - BM) * rowb + stage * 128;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsW, (__attribute__((address_space(3))) void*)((char*)smem + buf * STAGE + g * 1024),
                                                             16, lvoff, soff, 0, 0);
                }
            };
#pragma unroll
            for (int s = 0; s < NBUF - 1; ++s) issue(s < nq ? s : nq - 1, s);
            int nb = NBUF - 1;                                      // slot of stage s + NBUF - 1
            for (int s = 0; s < nq; ++s) {
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NBUF - 2) * PL) : "memory");
                issue(s + NBUF - 1 < nq ? s + NBUF - 1 : nq - 1, nb);
                nb = nb == NBUF - 1 ? 0 : nb + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            return;
        }
    }
    // lane -> (row within the instruction's 8, physical piece); logical piece = physical ^ ((stage row >> 1) & 7), stage row = 8g + r_in,
    // g = i * NWAVE + w has the parity of w (NWAVE is even): ((8g + r_in) >> 1) & 7 = (4 (w & 1) + (r_in >> 1)) & 7
    const int r_in = lane >> 3, pp = lane & 7;
    const int dvoff = r_in * rowb + ((pp ^ ((4 * (wave & 1) + (r_in >> 1)) & 7)) << 4);
    auto dma = [&](int i, int stage, int buf) {                     // i, buf: compile-time after unrolling; stage: uniform
        const int g = i * NWAVE + wave;
        const bool isA = g < BM / 8;                                // uniform
#ifdef H2P_EXP_NODMA
        const int soff = 0x7ffffff0;                                  // out of range: the DMA writes zeros, no memory traffic
#else
        const int soff = (isA ? 8 * g : 8 * g - BM) * rowb + stage * 128;
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsW, (__attribute__((address_space(3))) void*)((char*)smem + buf * STAGE + g * 1024),
                                                 16, dvoff, soff, 0, 0);
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
    f32x16 tot[FOLD ? 2 : 1][FOLD ? TN : 1];                        // FOLD: running sum of the finished slices
    const int per = FOLD ? nq / (d.ksplit > 1 ? d.ksplit : 1) : nq;  // stages per K slice
    int in_slice = 0, slice = 0;
    auto fold = [&]() __attribute__((always_inline)) {
        if constexpr (FOLD) {
            if (++in_slice < per) return;
            in_slice = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // the slice's partial sum exactly as the parallel form stores it: fma(accX, 2^-11, accH), then the epilogue's
                        // "+ bias" with no bias (x + 0.0f: turns -0 into +0)
                        const float pr = __builtin_fmaf(accX[a][b][r], 1.0f / 2048.0f, accH[a][b][r]) + 0.0f;
                        tot[a][b][r] = slice == 0 ? pr : tot[a][b][r] + pr;
                        accH[a][b][r] = 0.f; accX[a][b][r] = 0.f;
                    }
            ++slice;
        }
    };

    // prologue: stages 0 .. NBUF-2 in flight (a stage index beyond the last re-reads the last stage into a free buffer: every iteration
    // then issues exactly NI instructions, which keeps the vmcnt arithmetic uniform)
    if constexpr (NLOAD == 0) {
#pragma unroll
        for (int s = 0; s < NBUF - 1; ++s)
#pragma unroll
            for (int i = 0; i < NI; ++i) dma(i, s < nq ? s : nq - 1, s);
    }

    f16x8 af[2][2][2], bf[2][TN][2];                                // [t][a | b][plane]
    auto stage_body = [&](auto buf_tag, int s) {
        constexpr int BUF = decltype(buf_tag)::value, NXT = (BUF + NBUF - 1) % NBUF;
        // my DMA instructions of stage s have landed once at most (NBUF - 2) later stages' are outstanding; after the barrier everyone's
        // have, and everyone is done reading the buffer of stage s - 1 (= the one stage s + NBUF - 1 goes to)
        if constexpr (NLOAD > 0) asm volatile("s_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NBUF - 2) * NI) : "memory");
        LVAE_FENCE();
        unsigned aa[4], ba[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) { aa[pt] = a_base[pt] + BUF * STAGE; ba[pt] = b_base[pt] + BUF * STAGE; }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                H2P_DSR(af[tt][a][0], aa[0 + tt], a * 4096);
                H2P_DSR(af[tt][a][1], aa[2 + tt], a * 4096);
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                H2P_DSR(bf[tt][b][0], ba[0 + tt], b * 4096);
                H2P_DSR(bf[tt][b][1], ba[2 + tt], b * 4096);
            }
        }
        const int sn = s + NBUF - 1 < nq ? s + NBUF - 1 : nq - 1;
        int issued = 0;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            // fragments of step tt have landed when at most the (4 + 2 TN) reads of step 1 are outstanding
            if (tt == 0) {
                if constexpr (TN == 2) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(af[0][0][0]), "+v"(af[0][0][1]), "+v"(af[0][1][0]), "+v"(af[0][1][1]),
                                                    "+v"(bf[0][0][0]), "+v"(bf[0][0][1]), "+v"(bf[0][1][0]), "+v"(bf[0][1][1]));
                else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[0][0][0]), "+v"(af[0][0][1]), "+v"(af[0][1][0]), "+v"(af[0][1][1]),
                                  "+v"(bf[0][0][0]), "+v"(bf[0][0][1]));
            } else {
                if constexpr (TN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0][0]), "+v"(af[1][0][1]), "+v"(af[1][1][0]), "+v"(af[1][1][1]),
                                                    "+v"(bf[1][0][0]), "+v"(bf[1][0][1]), "+v"(bf[1][1][0]), "+v"(bf[1][1][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0][0]), "+v"(af[1][0][1]), "+v"(af[1][1][0]), "+v"(af[1][1][1]),
                                  "+v"(bf[1][0][0]), "+v"(bf[1][0][1]));
            }
            LVAE_FENCE();
#pragma unroll
            for (int b = 0; b < TN; ++b) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j == 0) {
                        accX[0][b] = H2P_MFMA(af[tt][0][1], bf[tt][b][0], accX[0][b]);
                        accX[1][b] = H2P_MFMA(af[tt][1][1], bf[tt][b][0], accX[1][b]);
                    } else if (j == 1) {
                        accX[0][b] = H2P_MFMA(af[tt][0][0], bf[tt][b][1], accX[0][b]);
                        accX[1][b] = H2P_MFMA(af[tt][1][0], bf[tt][b][1], accX[1][b]);
                    } else {
                        accH[0][b] = H2P_MFMA(af[tt][0][0], bf[tt][b][0], accH[0][b]);
                        accH[1][b] = H2P_MFMA(af[tt][1][0], bf[tt][b][0], accH[1][b]);
                    }
                    // one DMA instruction of stage s + NBUF - 1 behind each MFMA pair until all NI are out
                    if constexpr (NLOAD == 0) { if (issued < NI) { dma(issued, sn, NXT); ++issued; } }
                    LVAE_FENCE();
                }
            }
        }
        static_assert(NI <= 6 * TN, "DMA instructions per stage must fit the MFMA pairs of a stage");
    };
    for (int s = 0; s < nq; s += NBUF) {
        stage_body(std::integral_constant<int, 0>{}, s);
        fold();
        if (s + 1 < nq) { stage_body(std::integral_constant<int, 1 % NBUF>{}, s + 1); fold(); }
        if (NBUF > 2 && s + 2 < nq) { stage_body(std::integral_constant<int, 2 % NBUF>{}, s + 2); fold(); }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // trailing (redundant) DMAs have landed before LDS is reused / freed
    if constexpr (FOLD) {
#ifndef H2P_EXP_GENERIC_EPI
        if (d.bias && n0 + BN <= d.N && !((d.ldo | d.ldres) & 3)) {          // uniform: a straight-line form of the tail below
            if (d.epi == LVAE_EPI_BIAS_GELU && d.out_h2 && !(d.ldo & 31)) {
                h2p_epilogue_fast<TN, LVAE_EPI_BIAS_GELU, true, H2P_FULL_LINE, true>(d, tot, m0, n0, rows_a, wave_m, wave_n, li, lh);
                return;
            }
            if (d.epi == LVAE_EPI_GAMMA_RES && !d.out_h2) {
                h2p_epilogue_fast<TN, LVAE_EPI_GAMMA_RES, false, false, true>(d, tot, m0, n0, rows_a, wave_m, wave_n, li, lh);
                return;
            }
            if (d.epi == LVAE_EPI_RES && !d.out_h2) {
                h2p_epilogue_fast<TN, LVAE_EPI_RES, false, false, true>(d, tot, m0, n0, rows_a, wave_m, wave_n, li, lh);
                return;
            }
        }
#endif
        // the reduce kernel's tail on this tile: 4 consecutive columns of one row per lane (quad transpose), then its epilogue function
        const int lj = li & 3;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = m0 + (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    float v0 = tot[a][b][4 * g + 0], v1 = tot[a][b][4 * g + 1], v2 = tot[a][b][4 * g + 2], v3 = tot[a][b][4 * g + 3];
                    quad_transpose(v0, v1, v2, v3, lj);
                    const int c4 = n0 + (wave_n * TN + b) * 32 + (li & ~3);
                    if (row < d.M && c4 < d.N) splitk_epilogue_store<true>(d, row, c4, (f32x4){v0, v1, v2, v3});
                }
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) accH[a][b][r] = __builtin_fmaf(accX[a][b][r], 1.0f / 2048.0f, accH[a][b][r]);
#ifdef H2P_EXP_NOEPI
    if (accH[0][0][0] == 123.456f) d.out[0] = accH[0][0][1] + accH[1][TN - 1][3];
#else
#ifndef H2P_EXP_GENERIC_EPI
    if (d.store == LVAE_ST_ROWMAJOR && n0 + BN <= d.N && !((d.ldo | d.ldres) & 3)) {          // uniform: one of the straight-line forms
        if (d.epi == LVAE_EPI_BIAS_GELU && d.out_h2 && !(d.ldo & 31)) {
            h2p_epilogue_fast<TN, LVAE_EPI_BIAS_GELU, true, H2P_FULL_LINE>(d, accH, m0, n0, rows_a, wave_m, wave_n, li, lh);
            return;
        }
        if (d.epi == LVAE_EPI_GAMMA_RES && !d.out_h2) {
            h2p_epilogue_fast<TN, LVAE_EPI_GAMMA_RES, false, false>(d, accH, m0, n0, rows_a, wave_m, wave_n, li, lh);
            return;
        }
        if (d.epi == LVAE_EPI_RES && !d.out_h2) {
            h2p_epilogue_fast<TN, LVAE_EPI_RES, false, false>(d, accH, m0, n0, rows_a, wave_m, wave_n, li, lh);
            return;
        }
    }
#endif
    gemm_finish<C>(d, accH, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
#endif
}

template <int WM, int TN, int NBUF, bool FOLD = false, int NLOAD = 0>
int launch_h2p(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * TN, LDS = NBUF * (BM + BN) * 128;
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)gemm_h2p_kernel<WM, TN, NBUF, FOLD, NLOAD>, LDS)) return ae;
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
    static int lds_pad = 0, stagger = 0;
#ifdef LVAE_EXPERIMENTAL_BUILD           // knobs of the round-3 studies (tools/build_exp.sh copies only; docs/MEASUREMENT_HISTORY.md 5c): extra dynamic LDS (forces
    static bool env_read = false;        // one 128-row workgroup per CU) and the phase stagger (measured 0 ... -5 % on the model's shapes)
    if (!env_read) {
        const char* e = getenv("LVAE_H2P_LDSPAD"); lds_pad = e ? atoi(e) : 0;
        e = getenv("LVAE_H2P_STAGGER"); stagger = e ? atoi(e) : 0;
        env_read = true;
    }
    if (lds_pad > 0) (void)hipFuncSetAttribute((const void*)gemm_h2p_kernel<WM, TN, NBUF, FOLD, NLOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS + lds_pad);
#endif
    hipLaunchKernelGGL((gemm_h2p_kernel<WM, TN, NBUF, FOLD, NLOAD>), dim3(n_tiles), dim3(128 * WM + 64 * NLOAD), LDS + lds_pad, st, *d, tiles_n, n_tiles, stagger);
    return (int)hipGetLastError();
}

}  // namespace

// Entry point for gemm_f32.hip's dispatcher (prec 4, a_h2 = 1).  force: 0 = choose; 10 * WM + TN = that tile (tuning hook LVAE_H2P_TILE:
// 42 = 256 x 128, 41 = 256 x 64, 22 = 128 x 128, 21 = 128 x 64).  Every choice gives the same bits.
#ifdef LVAE_EXP_H2PP
int lvae_gemm_h2pp_launch(const lvae_gemm_desc* d, hipStream_t st, int tn);        // gemm_h2pp.hip: the persistent-form study (force = 92 / 91)
#endif

#ifdef LVAE_EXP_H2E
int lvae_gemm_h2e_try(const lvae_gemm_desc* d, hipStream_t st, int* rc);          // gemm_h2e.hip: the epilogue-interleaved study (force = 51)
#endif

#ifdef LVAE_EXP_H2Q
int lvae_gemm_h2q_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc);          // gemm_h2q.hip: the loader-wave study (force = 61)
#endif

int lvae_gemm_h2p_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc) {
    if (d->prec != 4 || !d->a_h2 || d->a_mode != LVAE_A_PLAIN || d->K1 != 0 || d->K0 != d->K || (d->K & 31) || d->lda0 != d->K ||
        d->ldw != d->K || d->a_gelu || (long)256 * d->K * 4 > 0x7fffffffL)
        return 0;
    const int M = d->M, N = d->N;
    if (d->ksplit > 1) {
        // serial split-K (FOLD): S slices of whole 32-deep stages, row-major fp32 / pre-split output with 16-B rows (what the reduce
        // kernel's epilogue function stores), 128 x 64 tiles (the running sum lives beside two accumulator sets)
        if ((d->K / 32) % d->ksplit || d->store != LVAE_ST_ROWMAJOR || (N & 3) || (d->ldo & 3) || (d->ldres & 3)) return 0;
        // up to one workgroup per CU: eight loader waves beside the four compute waves (above); more tiles: two workgroups per CU hide
        // each other's DMA issue as before (tuning hook of tools/r6_fold_loaders.sh: LVAE_FOLD_LOADERS=0 / 1)
#ifdef LVAE_EXPERIMENTAL_BUILD
        static const int force = getenv("LVAE_FOLD_LOADERS") ? atoi(getenv("LVAE_FOLD_LOADERS")) : -1;
#else
        constexpr int force = -1;                                   // (the product library's launch paths read no environment)
#endif
        const int tiles = ((M + 127) / 128) * ((N + 63) / 64);
        const bool loaders = force >= 0 ? force != 0 : tiles <= lvae_cu_count();
        *rc = loaders ? launch_h2p<2, 1, 3, true, 8>(d, st) : launch_h2p<2, 1, 3, true>(d, st);
        return 1;
    }
    int sel = force;
#ifdef LVAE_EXP_H2Q
    if (sel == 61) {
        if (lvae_gemm_h2q_try(d, st, 1, rc)) return 1;
        sel = 0;
    }
#endif
#ifdef LVAE_EXP_H2E
    if (sel == 51) {
        if (lvae_gemm_h2e_try(d, st, rc)) return 1;
        sel = 0;
    }
#endif
#ifdef LVAE_EXP_H2PP
    if ((sel == 92 || sel == 91) && d->K >= 128 && !(d->K & 63)) { *rc = lvae_gemm_h2pp_launch(d, st, sel - 90); return 1; }
#endif
    if (sel != 42 && sel != 41 && sel != 22 && sel != 21 && sel != 23) {
        // Tile by measurement (profiles/r03_gemm_h2p_tile_sweep.txt: every MLP shape of the model at batch 4 and 8 under each tile):
        // least padded width first (N = 192: three 64-wide tiles, not two 128-wide); 64-wide: 128 x 64 everywhere; 128-wide: 128 x 64
        // while 128 x 128 tiles would be fewer than ~4 per CU (the mid-size launches of one pipeline group: more, smaller workgroups
        // fill the chip and overlap their epilogues), 256 x 128 only for the largest (stride-4, batch 8) launches
        const int tn = (N % 128 == 0 || ((N + 127) / 128) * 128 - N < ((N + 63) / 64) * 64 - N + 1) ? 2 : 1;
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        sel = tn == 1 ? 21 : (t128 < 1024 ? 21 : (t128 >= 4096 ? 42 : 22));
        // Round 5: the 128 x 64 tile with TWO ring slots (48 KB of LDS: three workgroups per CU instead of two; its 164 registers allow
        // three waves per SIMD anyway).  One stage less of DMA look-ahead against a third workgroup whose epilogue can run under the
        // other two's main loops: -2 ... -7 % where the K loop is short and the tiles are many, +5 ... +60 % on the few-tile long-K
        // layers (profiles/r05_gemm_h2p_three_workgroups_per_cu.txt) -- taken for K <= 512 with at least two tiles per slot-pair
        const long t64 = (long)((M + 127) / 128) * ((N + 63) / 64);
        if (d->K <= 512 && t64 >= 512 && (sel == 21 || (sel == 22 && t128 < 2048))) sel = 23;
    }
    switch (sel) {
        case 42: *rc = launch_h2p<4, 2, 3>(d, st); break;
        case 41: *rc = launch_h2p<4, 1, 3>(d, st); break;
        case 22: *rc = launch_h2p<2, 2, 2>(d, st); break;
        case 23: *rc = launch_h2p<2, 1, 2>(d, st); break;      // 128 x 64, two ring slots (48 KB): three workgroups per CU

        default: *rc = launch_h2p<2, 1, 3>(d, st); break;
    }
    return 1;
}
