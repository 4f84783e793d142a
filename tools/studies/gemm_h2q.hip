// gemm_h2q.hip -- gemm_h2p's 128 x 64-tile GEMM (both operands pre-split H2K32, f16x2 arithmetic, the same bits) as a PERSISTENT kernel with
// a LOADER WAVE: four compute waves + one wave that does nothing but issue the LDS-DMA instructions and wait for them.
//
// Why.  A gemm_h2p launch is main loop + epilogue per workgroup, and the two add up (profiles/r04_gemm_h2p_ablations.txt: 49152 x 768 x 384
// main loop alone 52 us; + plain fp32 stores of the 151 MB result: 94 us): a workgroup's slot is busy until its stores have drained, and
// the chip drains stores at ~3.6 TB/s.  A persistent workgroup could compute its next tile while the stores of the last one drain -- but
// gfx950 has ONE vmcnt for loads and stores: a wave that has stores in flight cannot tell when its DMA loads have landed without
// waiting for the stores as well (mlp_h2c.hip pays exactly that at the start of every tile).  So the roles are split by WAVE:
//   * the loader wave (wave 4) walks the workgroup's stages -- tile after tile, the ring of three 24 KB slots running flat across tile
//     boundaries -- and per stage: waits until its DMAs of that stage have landed (counted vmcnt: it has only loads in flight), joins
//     the workgroup barrier, then issues the 24 DMA instructions of the stage two ahead into the slot the barrier just freed;
//   * the compute waves never wait on vmcnt for the ring: per stage the barrier, fragment reads (lgkmcnt), 12 MFMAs; per tile the
//     fused epilogue, whose stores are simply left behind -- they drain while the next tile's stages are computed.
// Two such workgroups (5 waves, 72 KB) per CU; 164 registers per wave leave room for three waves per SIMD.
// Arithmetic: gemm_h2p_kernel<2, 1, 3>'s fragment layout, MFMA order and fold, gemm_epilogue's operations in its order -- every output bit equal
// (tools/h2q_equal.py: 0 words differ on ten shapes).
// STUDY, NOT PART OF liblvae_hip.so (like gemm_h2e.hip / gemm_h2pp.hip): it lost.  EXTRA_SRC=gemm_h2q.hip tools/build_exp.sh h2q gemm_h2p.hip
// -DLVAE_EXP_H2Q links it behind d.cfg = 61.  Measured (profiles/r04_gemm_h2q_loader_wave_study.txt): 5 - 12 % SLOWER than gemm_h2p's own
// 128 x 64 tile on every MLP shape (49152 x 768 x 384 GELU 158-165 us against 141-147; bias-only 138-146 against 132-144), although the
// compute waves' instruction stream has no vmcnt wait left except one per tile in front of its stores.  What that says: the result
// stores do not cost a launch their LATENCY (which this form hides) but their BANDWIDTH -- the chip drains stores at ~3.6 TB/s (151 MB of
// hidden map: 42 us), a CU's vector-memory pipe is one in-order queue, and while it is backed up with one workgroup's stores the other
// workgroup's LDS-DMA loads wait behind them: main loop and store phase add up whoever issues what.  Only not writing the map helps
// (mlp_h2c.hip).
#include "../../lossy-vae_amd/csrc/gemm_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)
#define H2Q_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define H2Q_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

constexpr int Q_BM = 128, Q_BN = 64, Q_ROWS = Q_BM + Q_BN, Q_STAGE = Q_ROWS * 128, Q_NBUF = 3, Q_NG = Q_ROWS / 8;

__global__ __launch_bounds__(320, 2) void gemm_h2q_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    using C = Cfg<2, 2, 2, 1, 1, 32>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = d.K / 32;
    const int rowb = d.K * 4;                                       // bytes of one H2K32 row (A and W alike)
    // tiles of this workgroup: linear index blockIdx.x + k * gridDim.x, mapped so that consecutive tiles (n fastest) stay on one XCD
    // (gridDim.x is a multiple of 8: the linear index keeps the workgroup's XCD)
    const int n_my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_of = [&](int k) -> int {
        const int b = blockIdx.x + k * gridDim.x;
        const int q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    };
    const int total = n_my * nq;                                    // stages of this workgroup

    if (wave == 4) {
        // ================================================================== loader wave
        // instruction g of a stage covers its rows 8g .. 8g + 7 (A rows first); lane -> (row within the 8, physical 16-B piece); logical
        // piece = physical ^ ((stage row >> 1) & 7), stage row = 8g + r_in: ((8g + r_in) >> 1) & 7 = (4 (g & 1) + (r_in >> 1)) & 7
        const int r_in = lane >> 3, pp = lane & 7;
        int dv[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) dv[par] = r_in * rowb + ((pp ^ ((4 * par + (r_in >> 1)) & 7)) << 4);
        int k = 0, s = 0, slot = 0;                                   // the stage to issue next: tile k, stage s, into `slot`
        __amdgpu_buffer_rsrc_t rsA, rsW;
        auto open_tile = [&]() {
            if (k < n_my) {
                const int t = tile_of(k), tm = t / tiles_n, tn = t - tm * tiles_n, m0 = tm * Q_BM, n0 = tn * Q_BN;
                const int rows_a = (d.M - m0) < Q_BM ? (d.M - m0) : Q_BM, rows_w = (d.N - n0) < Q_BN ? (d.N - n0) : Q_BN;
                rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.A0 + (long)m0 * rowb), 0, rows_a * rowb, 0x00020000);
                rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.Wt16 + (long)n0 * rowb), 0, rows_w * rowb, 0x00020000);
            } else {
                // beyond the last tile: the same number of instructions (the vmcnt arithmetic stays uniform), every lane out of range --
                // zeros into a slot nobody reads, no memory traffic
                rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A0, 0, 0, 0x00020000);
                rsW = rsA;
            }
        };
        auto issue = [&]() {
            char* base = (char*)smem + slot * Q_STAGE;
#pragma unroll
            for (int g = 0; g < Q_NG; ++g) {
                const bool isA = g < Q_BM / 8;
                const int soff = (isA ? 8 * g : 8 * g - Q_BM) * rowb + s * 128;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsW, (__attribute__((address_space(3))) void*)(base + g * 1024), 16, dv[g & 1], soff, 0, 0);
            }
            slot = slot + 1 == Q_NBUF ? 0 : slot + 1;
            if (++s == nq) { s = 0; ++k; open_tile(); }
        };
        open_tile();
#pragma unroll
        for (int i = 0; i < Q_NBUF - 1; ++i) issue();                 // stages 0 .. NBUF - 2 in flight
        for (int gs = 0; gs < total; ++gs) {
            // stage gs has landed once at most the NBUF - 2 later stages' instructions are outstanding; after the barrier the compute waves
            // read it, and everyone is done with the slot of stage gs - 1 = the one stage gs + NBUF - 1 goes to
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((Q_NBUF - 2) * Q_NG) : "memory");
            issue();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ====================================================================== compute waves (gemm_h2p_kernel<2, 1, 3>'s wave tile: 64 x 32)
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // fragment side: piece (plane p, k16 step t, lane half lh) = 4p + 2t + lh, read at ((piece ^ x) << 4) of the lane's row
    const int xr = (li >> 1) & 7;
    unsigned a_base[4], b_base[4];                                  // [2p + t]: byte address inside a stage
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int piece = 4 * (pt >> 1) + 2 * (pt & 1) + lh;
        const unsigned o = (unsigned)((piece ^ xr) << 4);
        a_base[pt] = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem) + (wave_m * 64 + li) * 128 + o;
        b_base[pt] = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem) + (Q_BM + wave_n * 32 + li) * 128 + o;
    }
    int slot = 0;
    for (int k = 0; k < n_my; ++k) {
        const int t = tile_of(k), tm = t / tiles_n, tn = t - tm * tiles_n, m0 = tm * Q_BM, n0 = tn * Q_BN;
        f32x16 accH[2][1], accX[2][1];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accH[a][0][r] = 0.f; accX[a][0][r] = 0.f; }
        for (int s = 0; s < nq; ++s) {
            asm volatile("s_barrier" ::: "memory");
            LVAE_FENCE();
            unsigned so = (unsigned)(slot * Q_STAGE);
            asm volatile("" : "+v"(so));
            slot = slot + 1 == Q_NBUF ? 0 : slot + 1;
            f16x8 af[2][2][2], bf[2][2];                            // [t][a][plane], [t][plane]
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    H2Q_DSR(af[tt][a][0], a_base[0 + tt] + so, a * 4096);
                    H2Q_DSR(af[tt][a][1], a_base[2 + tt] + so, a * 4096);
                }
                H2Q_DSR(bf[tt][0], b_base[0 + tt] + so, 0);
                H2Q_DSR(bf[tt][1], b_base[2 + tt] + so, 0);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                // fragments of step tt have landed when at most the six reads of step 1 are outstanding
                if (tt == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[0][0][0]), "+v"(af[0][0][1]), "+v"(af[0][1][0]), "+v"(af[0][1][1]), "+v"(bf[0][0]), "+v"(bf[0][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0][0]), "+v"(af[1][0][1]), "+v"(af[1][1][0]), "+v"(af[1][1][1]), "+v"(bf[1][0]), "+v"(bf[1][1]));
                LVAE_FENCE();
                accX[0][0] = H2Q_MFMA(af[tt][0][1], bf[tt][0], accX[0][0]);
                accX[1][0] = H2Q_MFMA(af[tt][1][1], bf[tt][0], accX[1][0]);
                accX[0][0] = H2Q_MFMA(af[tt][0][0], bf[tt][1], accX[0][0]);
                accX[1][0] = H2Q_MFMA(af[tt][1][0], bf[tt][1], accX[1][0]);
                accH[0][0] = H2Q_MFMA(af[tt][0][0], bf[tt][0], accH[0][0]);
                accH[1][0] = H2Q_MFMA(af[tt][1][0], bf[tt][0], accH[1][0]);
                LVAE_FENCE();
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) accH[a][0][r] = __builtin_fmaf(accX[a][0][r], 1.0f / 2048.0f, accH[a][0][r]);
        // ---- epilogue: gemm_epilogue's vector path (row-major output, 16-B rows: the host's rule for this kernel), operation for operation,
        // but with EVERY load of the tile up front and waited for before the first store -- the compiler keeps a load that some path never
        // consumes pending on its register, and would meet it at the next tile as a vmcnt(0) in front of the accumulator reset: a wait for
        // all the stores that are meant to drain under the next tile
        {
            const int lj = li & 3, epi = d.epi;
            const int colb = n0 + wave_n * 32, col = colb + li, cc = col < d.N ? col : 0;
            const int c4 = colb + (li & ~3);
            const bool cok4 = c4 < d.N;
            const int c4c = cok4 ? c4 : 0;
            const bool has_res = epi == LVAE_EPI_GAMMA_RES || epi == LVAE_EPI_RES;
            const float cbias = d.bias ? d.bias[cc] : 0.f;
            const float cgam = (epi == LVAE_EPI_GAMMA_RES) ? d.gamma[cc] : 1.f;
            f32x4 rv[2][4];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = m0 + (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
                    rv[a][g] = has_res ? *(const f32x4*)(d.res + (long)(row < d.M ? row : 0) * d.ldres + c4c) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (only this tile's loads and the PREVIOUS tile's stores -- a main loop old -- are in flight)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = m0 + (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
                    const bool rok = row < d.M;
                    const long obase = (long)(rok ? row : 0) * d.ldo;
                    float v0 = accH[a][0][4 * g + 0] + cbias, v1 = accH[a][0][4 * g + 1] + cbias;
                    float v2 = accH[a][0][4 * g + 2] + cbias, v3 = accH[a][0][4 * g + 3] + cbias;
                    if (epi == LVAE_EPI_BIAS_GELU) { gelu_erf2(v0, v1); gelu_erf2(v2, v3); }
                    else if (epi == LVAE_EPI_GAMMA_RES) { v0 *= cgam; v1 *= cgam; v2 *= cgam; v3 *= cgam; }
                    quad_transpose(v0, v1, v2, v3, lj);
                    if (rok && cok4) {
                        f32x4 o = {v0, v1, v2, v3};
                        if (has_res) { o[0] += rv[a][g][0]; o[1] += rv[a][g][1]; o[2] += rv[a][g][2]; o[3] += rv[a][g][3]; }
                        if (d.out_h2) {
                            unsigned h0, l0, h1, l1;
                            split_pair_h2(o[0], o[1], h0, l0);
                            split_pair_h2(o[2], o[3], h1, l1);
                            const u32x2_t hi2 = {h0, h1}, lo2 = {l0, l1};
                            char* q = (char*)d.out + (obase << 2) + ((c4c >> 5) << 7) + ((c4c & 31) << 1);
                            *(u32x2_t*)q = hi2;
                            *(u32x2_t*)(q + 64) = lo2;
                        } else {
                            *(f32x4*)(d.out + obase + c4c) = o;
                        }
                    }
                }
        }
    }
}

}  // namespace

// Entry point for gemm_h2p.hip's dispatcher.  -> 1 when this kernel takes the problem (*rc = launch status).
int lvae_gemm_h2q_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc) {
    if (d->prec != 4 || !d->a_h2 || d->a_mode != LVAE_A_PLAIN || d->K1 != 0 || d->K0 != d->K || (d->K & 31) || d->lda0 != d->K || d->ldw != d->K ||
        d->a_gelu || d->ksplit > 1 || (long)256 * d->K * 4 > 0x7fffffffL)
        return 0;
    // the epilogue here is gemm_epilogue's vector path only: row-major output, 16-B aligned rows
    if (d->store != LVAE_ST_ROWMAJOR || (d->N & 3) || (d->ldo & 3) || (d->ldres & 3)) return 0;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int tiles_m = (d->M + Q_BM - 1) / Q_BM, tiles_n = (d->N + Q_BN - 1) / Q_BN, n_tiles = tiles_m * tiles_n;
    const int slots = 2 * n_cu;
    if (!force && n_tiles < 3 * slots) return 0;               // fewer than three tiles per workgroup: nothing to overlap
    constexpr int LDS = Q_NBUF * Q_STAGE;
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)gemm_h2q_kernel, LDS)) { *rc = ae; return 1; }
    int grid = n_tiles < slots ? (n_tiles + 7) / 8 * 8 : slots;      // a multiple of 8 (tile_of keeps a workgroup on its XCD)
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(gemm_h2q_kernel, dim3(grid), dim3(320), LDS, st, *d, tiles_n, n_tiles);
    *rc = (int)hipGetLastError();
    return 1;
}
