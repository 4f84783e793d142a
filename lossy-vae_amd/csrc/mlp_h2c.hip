// mlp_h2c.hip -- the MLP of a ConvNeXt block as ONE persistent kernel (f16x2 arithmetic, pre-split operands):
//     out = x + gamma * ( fc2( gelu( fc1(y) + b1 ) ) + b2 )                      (lvae/models/common.py:131-132,154-158)
// for the stride-4 blocks: the encoder's seven with C = 192 / hidden = 384 (qarv/zoo.py:38-40) and the decoder's eight with C = 128 /
// hidden = 192 (qarv/zoo.py:86-87: the GPU tail of every decode -- after the last latent block nothing else is left to overlap with).  As
// two launches (gemm_h2p.hip) such a block writes and re-reads its hidden map through HBM (302 / 151 MB each way at batch 8: the two
// launches run at 2.3 / 3.4 TB/s and 146 / 164 TFLOP/s, profiles/r03_op_times_*), and every tile's GELU / split / store epilogue is
// serial with its short main loop (K = 192: six 32-deep stages).  Here a workgroup (8 waves, 128 rows) walks the hidden dimension in
// chunks of HC (128; 64 for hidden = 192) and keeps the chunk on the CU:
//   per chunk:  P[128 x HC] = y W1_c^T over K = C          (C/32 "F" stages: A rows + the chunk's W1 rows)
//               H = split(gelu(P + b1_c)) -> LDS           (HC/32 stages of 128 rows x 128 B: the stage layout of an A operand)
//               O[128 x C] += H W2_c^T over K = HC         (HC/32 "G" stages: the chunk's W2 columns, C x 128 B per stage)
//   per tile:   out = res + gamma * (O + b2)
// All global -> LDS traffic is LDS-DMA (`buffer_load ... lds`, whole 128-B lines, source-side swizzle: gemm_h2p.hip) through ONE ring of
// three slots that runs flat across F stages, G stages, chunks and TILES (the workgroup is persistent: the next tile's first two
// stages are in flight while this tile's epilogue stores drain), two stages ahead, one raw s_barrier + counted vmcnt per stage.
// Every (tile-relative) stage position is compile-time -- the whole tile is unrolled -- so ring slots, LDS offsets and the vmcnt
// allowances are immediates.  LDS: 3 x 32 KB ring + 64 KB hidden chunk = the CU's 160 KB (192 / 384); 3 x 24 + 32 KB (128 / 192).
// Arithmetic per element is the two-launch path's, operation for operation: per accumulator the MFMA sequence of gemm_h2p_kernel (k16
// steps ascending -- chunks and G stages ascend in the hidden index; X: a_lo' w_hi, a_hi w_lo'; H: a_hi w_hi), fma(accX, 2^-11, accH),
// gemm_epilogue's "+ bias -> gelu" / "+ bias, * gamma, + residual" with the same roundings (no contraction), the same split_pair_h2 --
// so every output bit equals fc2(fc1(.)) through gemm_h2p (tests/test_gpu_f16x2.py::test_mlp_h2c_equals_two_gemms,
// test_mlp_h2f_equals_two_gemms) and the host may use it for these block shapes at every batch size.
// (Round 3's form for 128 / 192 -- mlp_h2f_kernel: one tile per workgroup, A and ALL of W1 fetched first, five phases in series, waves
//  waiting 47 % of their cycles -- is superseded by the <128, 192, 64> instance: same bits, 141 -> 122 us at M = 196608,
//  profiles/r04_mlp_h2c_128x192_vs_mlp_h2f.txt.)
#include "gemm_common.h"

#include <type_traits>
#include <utility>

#if !defined(LVAE_EXPERIMENTAL_BUILD) && (defined(H2C_EXP_NOGELU) || defined(H2C_EXP_NOADMA) || defined(H2C_EXP_NOWDMA) || defined(H2C_EXP_NOMFMA) || \
    defined(H2C_EXP_NOEPI) || defined(H2C_EXP_NODSR) || defined(H2C_EXP_NOBAR) || defined(H2C_EXP_TRACE) || defined(H2C_EXP_NOARES))
#error "H2C_EXP_* ablations (wrong results by construction: they remove work to time what is left) need -DLVAE_EXPERIMENTAL_BUILD (tools/build_exp.sh)"
#endif
// H2C_EXP_TRACE: in-kernel timeline (s_memtime) of wave 0 of workgroup 0 on its SECOND tile, written behind the output rows
// (out + M * C floats; tools/microbench.py mlptrace allocates the room): slot 3P .. 3P + 2 = before the counted wait / before the barrier /
// behind the barrier of position P; 96 + 2 ch .. = GELU phase of chunk ch begins / ends; 104 .. 106 = epilogue begins / next tile's stages
// landed / stores issued
#ifdef H2C_EXP_TRACE
#define H2C_T(slot) do { if (trace_on) trace[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define H2C_T(slot) do { } while (0)
#endif
#ifdef H2C_EXP_NOMFMA
#define H2C_MFMA(a, b, c) (c)
#else
#define H2C_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef H2C_EXP_NODSR
#define H2C_DSR(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr))
#else
#define H2C_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int C_, int HID_, int HC_, int BM_, int NSUB_, int NBUF_, bool ARES_ = false>
struct H2C {
    // ARES (round 5; the <128, 192, 64> instance, which leaves 56 KB of the CU's LDS unused): the tile's A rows are RESIDENT -- fetched once
    // per tile into their own region (C / 32 stages of BM rows x 128 B) instead of once per hidden chunk into the ring, an F stage's ring
    // slot then holds the chunk's W1 rows only.  The next tile's A rows are requested during the current tile's last chunk, as soon as
    // their stage has had its last reader (H2C::nextra): no position of a tile waits for a cold fetch from HBM any more (position 2 did:
    // 3-5 us per tile, profiles/r05_mlp_h2c_128x192_ablations.txt), and a third of the kernel's DMA bytes is gone.
    static constexpr bool ARES = ARES_;
    // HC: hidden chunk (128; 64 where 128 does not divide the hidden width).  BM: rows per tile (128; 64 where a 128-row output tile does
    // not fit the registers: C = 384).  NSUB: a G stage holds the W2 rows of C / NSUB output columns (1; 3 for C = 384: 128 rows = 16 KB)
    static constexpr int C = C_, HID = HID_, BM = BM_, HC = HC_, NSUB = NSUB_;
    static constexpr int WMN = BM / 32, WNN = 8 / WMN;   // the eight waves as WMN x WNN wave tiles of 32 rows
    static constexpr int NCH = HID / HC;                 // hidden chunks per tile
    static constexpr int KS1 = C / 32, KS2 = HC / 32;    // F stages / k32 steps of fc2 per chunk
    static constexpr int KG = KS2 * NSUB;                // G stage positions per chunk (k32 step major, column part minor)
    static constexpr int PT = KS1 + KG;                  // stage positions per chunk
    static constexpr int NP = NCH * PT;                  // ... per tile
    static constexpr int NBF = HC / (32 * WNN);          // hidden column blocks of a wave in an F stage (wave tile 32 x HC / WNN)
    static constexpr int NBS = C / (NSUB * 32 * WNN);    // output column blocks of a wave in ONE G stage
    static constexpr int NB2 = NSUB * NBS;               // output column blocks of a wave
    static constexpr int NBM = NBF > NBS ? NBF : NBS;
    static constexpr int FROWS = ARES ? HC : BM + HC;    // rows of an F stage's ring slot
    static constexpr int SLOT = (FROWS > C / NSUB ? FROWS : C / NSUB) * 128, NBUF = NBUF_, RING = NBUF * SLOT;   // a stage: [BM A rows +] HC W1 rows | C / NSUB W2 rows
    static constexpr int HBYTES = KS2 * BM * 128;        // hidden chunk: HC / 32 stages of BM rows x 128 B
    static constexpr int ABYTES = ARES ? KS1 * BM * 128 : 0;   // resident A tile
    static constexpr int AOFF = RING + HBYTES;
    static constexpr int LDS = RING + HBYTES + ABYTES;
    static constexpr int LA = NBUF - 1;                  // DMA look-ahead in stage positions (3-slot ring: two; 4 slots where LDS has the room and the stages are short)
    static constexpr int NA = BM / 64;                   // DMA instruction rounds that cover the A rows (8 rows each, 8 waves)
    static constexpr int NW1 = HC / 64;                  // DMA instruction rounds that cover the W1 rows of a chunk
    static constexpr int NI_F = (BM + HC) / 64;          // DMA instructions per wave: F stage (A rows + W1 rows)
    static constexpr int NI_G = C / NSUB / 64;           // ... G stage (C / NSUB rows of W2)
    static_assert(HID % HC == 0 && BM % 64 == 0 && HC % (32 * WNN) == 0 && C % (NSUB * 32 * WNN) == 0 && (C / NSUB) % 64 == 0 && NP % NBUF == 0 && NBM <= 3, "shape");
    static_assert(HC % 64 == 0, "W1 rows in whole DMA rounds");
    static_assert(!ARES || (KS1 == 4 && NBUF_ == 3 && NCH >= 2), "resident A rows: the schedule of H2C::nextra is written for four A stages and a look-ahead of two");
    static_assert(LDS <= 160 * 1024, "LDS");
    static constexpr bool is_f(int p) { return (p % PT) < KS1; }
    // DMA instructions per wave of the group of tile-relative position p (issued during position p - LA).  ARES: an F stage's group is
    // the chunk's W1 rows only; the NEXT tile's A rows are extra instructions of the LAST chunk's stages (nextra): A stage 0 during the
    // position behind its last reader (the last chunk's F0), A stage 1 one later -- six to seven microseconds before anyone waits for
    // them --, A stages 2 / 3 at the very end of the tile's last position, so that the wait in front of the epilogue's stores can leave
    // exactly them outstanding (loads retire in order): they are first needed at the next tile's position 2
    static constexpr int ni(int p) {
        const int q = p % NP;
        if (!is_f(q)) return NI_G;
        return ARES ? NW1 : NI_F;
    }
    static constexpr int nextra(int p) {
        if (!ARES || p < 0) return 0;
        const int q = p % NP;
        return (q == (NCH - 1) * PT + 1 || q == (NCH - 1) * PT + 2) ? NA : (q == NP - 1 ? 2 * NA : 0);
    }
    static constexpr int nextra_between(int p) { int n = 0; for (int x = p - LA; x < p; ++x) n += nextra(x); return n; }   // extras younger than position p's group
    static constexpr int ni_between(int p) { int n = 0; for (int x = p + 1; x < p + LA; ++x) n += ni(x); return n; }   // DMA instructions younger than position p's
};

// wait for a k16 step's fragment reads (a_[2], w_[NB][2]) -- the asm names them as operands so that no MFMA that reads them moves above it
template <int NB>
__device__ __forceinline__ void h2c_frags_landed(f16x8 (&a)[2], f16x8 (*w)[2]) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NB == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(w[0][0]), "+v"(w[0][1]));
    else if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]));
#endif
}

template <int C_, int HID_, int HC_, int BM_, int NSUB_, int NBUF_, bool ARES_> // (integer parameters: a kernel template over a type of the anonymous namespace gets no host stub symbol)
__global__ __launch_bounds__(512, 1) void mlp_h2c_kernel(const lvae_mlp_desc d, int n_tiles) {
    // (device pass only: hipcc's HOST pass cannot instantiate the generic lambdas below -- the kernel template then silently drops out
    //  of overload resolution and no launch stub is emitted; the host needs nothing but the stub)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
    using S = H2C<C_, HID_, HC_, BM_, NSUB_, NBUF_, ARES_>;
    constexpr int LA = S::LA;
    constexpr bool ARES = S::ARES;
    constexpr int C = S::C, HID = S::HID, BM = S::BM, KS1 = S::KS1, KS2 = S::KS2, PT = S::PT, NP = S::NP, NB2 = S::NB2, NBF = S::NBF, NBS = S::NBS, NSUB = S::NSUB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / S::WNN, wn = wave % S::WNN;
    // first column of this wave's output block ob (ob = column part * NBS + block within the part)
    auto col_of = [&](int ob) { return (ob / NBS) * (C / NSUB) + wn * 32 * NBS + 32 * (ob % NBS); };
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem);

    // per-column parameters of this lane's output columns (requested before the DMA queue fills: loads retire in order)
    float b2v[NB2], gmv[NB2], b1v[S::NCH][NBF];
#pragma unroll
    for (int b = 0; b < NB2; ++b) { b2v[b] = d.b2[col_of(b) + li]; gmv[b] = d.gamma[col_of(b) + li]; }
#pragma unroll
    for (int ch = 0; ch < S::NCH; ++ch)
#pragma unroll
        for (int b = 0; b < NBF; ++b) b1v[ch][b] = d.b1[ch * S::HC + 32 * NBF * wn + 32 * b + li];   // fc1 bias of this lane's hidden columns, every chunk

    // ---- DMA side: instruction g of an operand block covers its rows 8g .. 8g + 7 (one 128-B line each); wave w issues g = i * 8 + w,
    // so g has the parity of w and the source permutation ((row >> 1) & 7 = (4 (w & 1) + (r_in >> 1)) & 7) is a per-lane constant
    const int r_in = lane >> 3, pp = lane & 7;
    const int perm = (pp ^ ((4 * (wave & 1) + (r_in >> 1)) & 7)) << 4;
    const int dvA = r_in * (C * 4) + perm;               // A / W1 rows are C * 4 bytes (H2K32)
    const int dvW2 = r_in * (HID * 4) + perm;            // W2 rows are HID * 4 bytes
#ifdef H2C_EXP_NOWDMA
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w1, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w2, 0, 0, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w1, 0, HID * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w2, 0, C * HID * 4, 0x00020000);
#endif
    // issue the DMA instructions of tile-relative position P (compile time) of the tile whose A rows start at `abase` (`arows` valid rows;
    // 0 rows = nothing to fetch: every lane out of range, the slot is zero-filled -- keeps the instruction count, hence vmcnt, uniform)
    auto dma_pos = [&](auto ptag, int i, const char* abase, int arows) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value % NP, CH = P / PT, Q = P % PT, SL = P % S::NBUF;
        char* slot = (char*)smem + SL * S::SLOT;
        // (opaque copy: without it hipcc hoists the scalar offsets of all NP x NI instructions out of the tile loop and spills them)
        int wv = wave;
        asm volatile("" : "+s"(wv));
        const int g = i * 8 + wv;
        if constexpr (Q < KS1 && ARES) {                 // F stage Q of chunk CH, resident A rows: the chunk's W1 rows, k32 index Q
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (__attribute__((address_space(3))) void*)(slot + g * 1024), 16, dvA,
                                                     (CH * S::HC + 8 * g) * (C * 4) + Q * 128, 0, 0);
        } else if constexpr (Q < KS1) {                  // F stage Q of chunk CH: A rows 0 .. BM - 1 | W1 rows CH * HC .. + HC - 1, k32 index Q
            if (i < S::NA) {
#ifdef H2C_EXP_NOADMA
                arows = 0;
#endif
                const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, arows * (C * 4), 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(slot + g * 1024), 16, dvA, 8 * g * (C * 4) + Q * 128, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (__attribute__((address_space(3))) void*)(slot + g * 1024), 16, dvA,
                                                         (CH * S::HC + 8 * (g - BM / 8)) * (C * 4) + Q * 128, 0, 0);
            }
        } else {                                         // G stage: W2 rows of column part (Q - KS1) % NSUB, k32 index CH * KS2 + (Q - KS1) / NSUB of the hidden dimension
            constexpr int KK = (Q - KS1) / NSUB, SUB = (Q - KS1) % NSUB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (__attribute__((address_space(3))) void*)(slot + g * 1024), 16, dvW2,
                                                     (SUB * (C / NSUB) + 8 * g) * (HID * 4) + (CH * KS2 + KK) * 128, 0, 0);
        }
    };

    // ARES: round `ia` (8 rows per wave) of stage `stg` of a tile's resident A rows
    auto dma_arow = [&](int stg, int ia, const char* abase, int arows) __attribute__((always_inline)) {
        int wv = wave;
        asm volatile("" : "+s"(wv));
        const int ga = ia * 8 + wv;
#ifdef H2C_EXP_NOADMA
        arows = 0;
#endif
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, arows * (C * 4), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)((char*)smem + S::AOFF + stg * (BM * 128) + ga * 1024), 16, dvA,
                                                 8 * ga * (C * 4) + stg * 128, 0, 0);
    };
    // extra instruction e of tile-relative position P (H2C::nextra): the NEXT tile's A rows
    auto dma_extra = [&](auto ptag, int e, const char* abase, int arows) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value % NP;
        constexpr int STG0 = P == NP - 1 ? 2 : P - ((S::NCH - 1) * PT + 1);       // first A stage this position fetches
        dma_arow(STG0 + e / S::NA, e % S::NA, abase, arows);
    };

    // fragment addresses: piece (plane p, k16 step t, lane half) = 4p + 2t + lh at ((piece ^ x) << 4) of the lane's row; the hidden
    // chunk uses rot3 of the row permutation: any bijection of (row >> 1) & 7 keeps the fragment reads conflict-free, and this one also
    // separates rows m and m + 2 in the ds_write_b64 pattern of the GELU phase (4 rows x 4 column groups per 16 lanes), which the plain
    // form maps to the same banks (two-way conflicts on 31 % of the LDS cycles of round 3's kernel: profiles/r03_pmc_mlp_h2f.txt)
    const int xr = (li >> 1) & 7, xh = ((xr << 1) & 7) | (xr >> 2);
    unsigned po[4], ph[4];                               // [2p + t]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        po[pt] = (unsigned)(((4 * (pt >> 1) + 2 * (pt & 1) + lh) ^ xr) << 4);
        ph[pt] = (unsigned)(((4 * (pt >> 1) + 2 * (pt & 1) + lh) ^ xh) << 4);
    }
    const unsigned a_row = lds0 + (ARES ? S::AOFF : 0) + (32 * wm + li) * 128;   // + slot: A rows of an F stage (ARES: + stage * BM * 128 of the resident tile)
    const unsigned w1_row = lds0 + (ARES ? 0 : BM * 128) + (32 * NBF * wn + li) * 128;     // + slot: W1 rows of an F stage
    const unsigned w2_row = lds0 + (wn * 32 * NBS + li) * 128;                // + slot: W2 rows of a G stage
    const unsigned h_row = lds0 + S::RING + (32 * wm + li) * 128;             // + g * 16 KB: hidden rows

    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const char* ybase = (const char*)d.y;
    auto tile_abase = [&](int t) { return ybase + (long)t * BM * (C * 4); };
    auto tile_rows = [&](int t) { const int r = d.M - t * BM; return t < n_tiles ? (r < BM ? r : BM) : 0; };

    // prologue: positions 0 and 1 of the first tile
    {
        const char* ab = tile_abase(tile);
        const int ar = tile_rows(tile);
        if constexpr (ARES) static_for<KS1 * S::NA>([&](auto e) { dma_arow(decltype(e)::value / S::NA, decltype(e)::value % S::NA, ab, ar); });   // (older than the groups below)
        static_for<LA>([&](auto pp) { static_for<S::ni(decltype(pp)::value)>([&](auto i) { dma_pos(pp, decltype(i)::value, ab, ar); }); });
    }

    f32x16 oH[NB2], oX[NB2], pH[NBF], pX[NBF];
    // fragments of a stage's SECOND k16 step are read in the middle of the stage and consumed at the start of the NEXT stage of the same
    // kind (fc1 / fc2), behind that stage's barrier: its first MFMAs are then ready the moment the barrier opens, and cover the latency
    // of its own first fragment reads (eight waves reading at once: ~200 cycles during which the matrix pipe used to idle, 30 times a tile)
    f16x8 ca[2], cw[S::NBM][2];                                             // carried: A planes, W blocks x planes
    f32x4 rv[4][NB2];                                                        // residual rows of this tile (requested four stages early)
    bool first = true;
#ifdef H2C_EXP_TRACE
    unsigned long long* trace = (unsigned long long*)(d.out + (long)d.M * C);
    int tile_no = 0;
#endif
    for (; tile < n_tiles; tile += gridDim.x) {
#ifdef H2C_EXP_TRACE
        const bool trace_on = blockIdx.x == 0 && tid == 0 && tile_no == 1;
        ++tile_no;
#endif
        const int m0 = tile * BM;
        const int nxt = tile + gridDim.x;
        const char* ab_cur = tile_abase(tile);
        const int ar_cur = tile_rows(tile);
        const char* ab_nxt = tile_abase(nxt < n_tiles ? nxt : tile);
        const int ar_nxt = tile_rows(nxt);
#pragma unroll
        for (int b = 0; b < NB2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { oH[b][r] = 0.f; oX[b][r] = 0.f; }

        static_for<NP>([&](auto ptag) {
            constexpr int P = decltype(ptag)::value, CH = P / PT, Q = P % PT, SL = P % S::NBUF;
            constexpr bool F = Q < KS1;
            constexpr bool RUN_FIRST = Q == 0 || Q == KS1, RUN_LAST = Q == KS1 - 1 || Q == PT - 1;
            constexpr int NB = F ? NBF : NBS;                                // W blocks of this stage's wave tile
            constexpr int KK = F ? 0 : (Q - KS1) / NSUB;                     // G: k32 step of the chunk, and the first output block of this stage's column part
            constexpr int OB = F ? 0 : ((Q - KS1) % NSUB) * NBS;
            constexpr int OBP = (F || Q == KS1) ? 0 : ((Q - KS1 - 1) % NSUB) * NBS;      // ... of the previous G stage (whose second k16 step runs here)
            constexpr int P2 = P + LA;                                       // the position whose DMAs are issued during this stage
            constexpr int NIG = S::ni(P2), NI2 = NIG + S::nextra(P);             // this stage's DMA instructions: the group of P2, then the extras of P
            const char* ab2 = P2 >= NP ? ab_nxt : ab_cur;
            const int ar2 = P2 >= NP ? ar_nxt : ar_cur;
            if constexpr (Q == 0) {
#pragma unroll
                for (int b = 0; b < NBF; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { pH[b][r] = 0.f; pX[b][r] = 0.f; }
            }
            // ---- my DMA instructions of this position have landed once only those of position P + 1 are outstanding (+ the residual
            // rows, requested behind the last chunk's GELU phase: younger than the DMAs of its first two G stages); after the barrier
            // everyone's have, and everyone is done with position P - 1 -- fragment reads included (lgkmcnt: the slot of position P - 1 is
            // the target of the DMAs issued below).  Positions 0 and 1 of a tile that follows another one were waited for before that
            // tile's epilogue stores.
            H2C_T(3 * P);
            if (P >= LA || first) {
                constexpr int ALLOW = S::ni_between(P) + S::nextra_between(P) + ((CH == S::NCH - 1 && Q >= KS1 && Q < KS1 + LA) ? 4 * NB2 : 0);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ALLOW) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            H2C_T(3 * P + 1);
#ifndef H2C_EXP_NOBAR
            asm volatile("s_barrier" ::: "memory");
#endif
            H2C_T(3 * P + 2);
            LVAE_FENCE();
            int issued = 0;
            // (opaque: the per-slot, per-piece addresses are recomputed per stage -- a few v_add in MFMA shadows -- instead of being hoisted
            //  out of the tile loop as ~30 loop-invariant registers, which hipcc then spills)
            unsigned aq = F ? (ARES ? a_row + Q * (BM * 128) : a_row + SL * S::SLOT) : h_row + KK * (BM * 128);
            unsigned wq = (F ? w1_row : w2_row) + SL * S::SLOT;
            asm volatile("" : "+v"(aq), "+v"(wq));
            const unsigned* pa = F ? po : ph;                                 // piece offsets of the A side (hidden chunk: rot3 permutation)
            // one k16 step's MFMAs (three groups: X += a_lo' w_hi | X += a_hi w_lo' | H += a_hi w_hi, per accumulator in gemm_h2p's order),
            // one DMA instruction of position P + 2 behind each group
            auto mfma_step = [&](const f16x8 (&a)[2], const f16x8 (*w)[2], auto obtag) __attribute__((always_inline)) {
                constexpr int O = decltype(obtag)::value;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if constexpr (F) {
                            if (j == 0) pX[b] = H2C_MFMA(a[1], w[b][0], pX[b]);
                            else if (j == 1) pX[b] = H2C_MFMA(a[0], w[b][1], pX[b]);
                            else pH[b] = H2C_MFMA(a[0], w[b][0], pH[b]);
                        } else {
                            if (j == 0) oX[O + b] = H2C_MFMA(a[1], w[b][0], oX[O + b]);
                            else if (j == 1) oX[O + b] = H2C_MFMA(a[0], w[b][1], oX[O + b]);
                            else oH[O + b] = H2C_MFMA(a[0], w[b][0], oH[O + b]);
                        }
                    }
                    if (issued < NIG) { dma_pos(std::integral_constant<int, P2>{}, issued, ab2, ar2); ++issued; }
                    else if (issued < NI2) { dma_extra(ptag, issued - NIG, ab_nxt, ar_nxt); ++issued; }
                    LVAE_FENCE();
                }
            };
            // first k16 step of this stage -> fresh registers
            f16x8 fa[2], fw[NB][2];
            H2C_DSR(fa[0], aq + pa[0], 0);
            H2C_DSR(fa[1], aq + pa[2], 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                H2C_DSR(fw[b][0], wq + po[0], b * 4096);
                H2C_DSR(fw[b][1], wq + po[2], b * 4096);
            }
            LVAE_FENCE();
            if constexpr (!RUN_FIRST) mfma_step(ca, cw, std::integral_constant<int, OBP>{});       // the previous stage's second step (fragments carried)
            h2c_frags_landed<NB>(fa, fw);
            LVAE_FENCE();
            // second k16 step -> the carried registers (their last readers, the MFMAs above, have been issued)
            H2C_DSR(ca[0], aq + pa[1], 0);
            H2C_DSR(ca[1], aq + pa[3], 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                H2C_DSR(cw[b][0], wq + po[1], b * 4096);
                H2C_DSR(cw[b][1], wq + po[3], b * 4096);
            }
            LVAE_FENCE();
            mfma_step(fa, fw, std::integral_constant<int, OB>{});
            if constexpr (RUN_LAST) {
                h2c_frags_landed<NB>(ca, cw);
                LVAE_FENCE();
                mfma_step(ca, cw, std::integral_constant<int, OB>{});
            }
            static_assert(S::NI_F <= 6 && S::NI_G <= 6 && (!S::ARES || S::NW1 + 2 * S::NA <= 6), "DMA instructions of a stage must fit behind the MFMA groups of two k16 steps");
#pragma unroll
            for (int i2 = 0; i2 < 6; ++i2)                                    // (a run's first stage has three groups only)
                if (issued < NIG) { dma_pos(std::integral_constant<int, P2>{}, issued, ab2, ar2); ++issued; }
                else if (issued < NI2) { dma_extra(ptag, issued - NIG, ab_nxt, ar_nxt); ++issued; }
            if constexpr (Q == KS1 - 1) {
                // ---- GELU phase: hidden chunk = split(gelu(P + b1)) -> LDS in the stage layout of an A operand (row m of stage
                // c / 32: 64 B hi | 64 B lo', 16-B pieces permuted).  After the quad transpose a lane holds 4 consecutive columns of a row.
                // (opaque lane coordinates: the 16 store addresses of this phase are recomputed here instead of living -- spilled --
                //  across the whole tile loop)
                int lio = li, lho = lh;
                asm volatile("" : "+v"(lio), "+v"(lho));
                const int lj = lio & 3;
                H2C_T(96 + 2 * CH);
#pragma unroll
                for (int b = 0; b < NBF; ++b) {
                    const int cs = NBF * wn + b;                              // hidden columns 32 NBF wn + 32 b .. + 31 of the chunk = stage cs
                    const int cc = lio & ~3;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v0 = __builtin_fmaf(pX[b][4 * g + 0], 1.0f / 2048.0f, pH[b][4 * g + 0]) + b1v[CH][b];
                        float v1 = __builtin_fmaf(pX[b][4 * g + 1], 1.0f / 2048.0f, pH[b][4 * g + 1]) + b1v[CH][b];
                        float v2 = __builtin_fmaf(pX[b][4 * g + 2], 1.0f / 2048.0f, pH[b][4 * g + 2]) + b1v[CH][b];
                        float v3 = __builtin_fmaf(pX[b][4 * g + 3], 1.0f / 2048.0f, pH[b][4 * g + 3]) + b1v[CH][b];
#ifdef H2C_EXP_NOGELU
                        unsigned h0 = __float_as_uint(v0), l0 = __float_as_uint(v1), h1 = __float_as_uint(v2), l1 = __float_as_uint(v3);
#else
                        gelu_erf4(v0, v1, v2, v3);
                        quad_transpose(v0, v1, v2, v3, lj);
                        unsigned h0, l0, h1, l1;
                        split_pair_h2(v0, v1, h0, l0);
                        split_pair_h2(v2, v3, h1, l1);
#endif
                        const int m = 32 * wm + 4 * lho + 8 * g + lj;
                        const int k = (m >> 1) & 7, x = ((k << 1) & 7) | (k >> 2);
                        const unsigned base = lds0 + S::RING + cs * (BM * 128) + m * 128 + ((cc & 7) << 1);
                        const u32x2_t hi2 = {h0, h1}, lo2 = {l0, l1};
                        asm volatile("ds_write_b64 %0, %1" ::"v"(base + ((((cc >> 3) + 0) ^ x) << 4)), "v"(hi2) : "memory");
                        asm volatile("ds_write_b64 %0, %1" ::"v"(base + ((((cc >> 3) + 4) ^ x) << 4)), "v"(lo2) : "memory");
                    }
                }
                H2C_T(97 + 2 * CH);
                if constexpr (CH == S::NCH - 1) {
                    // the tile's residual rows, requested now (the P accumulators are dead: their registers hold the 4 * NB2 vectors), four
                    // fc2 stages before the epilogue adds them -- their latency under load (2 - 4 us) used to be exposed once per tile
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = m0 + 32 * wm + 4 * lho + 8 * g + lj;
                        const int rbg = (row < d.M ? row : 0) * C;            // (M * C < 2^31: checked on the host)
#pragma unroll
                        for (int b = 0; b < NB2; ++b)
#ifdef H2C_EXP_NOEPI
                            rv[g][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#else
                            rv[g][b] = *(const f32x4*)(d.res + rbg + col_of(b) + (lio & ~3));
#endif
                    }
                }
            }
        });
        first = false;

        // ---- epilogue 2: out = res + gamma * (O + b2)   (gemm_epilogue's order: + bias, * gamma, transpose, + residual).  The next tile's
        // first two stages (in flight since the last two G stages) are waited for HERE, before the stores: a counted vmcnt behind a store
        // burst would make the next stages wait for the stores to drain.
        {
            int lio = li, lho = lh;
            asm volatile("" : "+v"(lio), "+v"(lho));
            const int lj = lio & 3;
            H2C_T(104);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int b = 0; b < NB2; ++b) {
                    float v0 = (__builtin_fmaf(oX[b][4 * g + 0], 1.0f / 2048.0f, oH[b][4 * g + 0]) + b2v[b]) * gmv[b];
                    float v1 = (__builtin_fmaf(oX[b][4 * g + 1], 1.0f / 2048.0f, oH[b][4 * g + 1]) + b2v[b]) * gmv[b];
                    float v2 = (__builtin_fmaf(oX[b][4 * g + 2], 1.0f / 2048.0f, oH[b][4 * g + 2]) + b2v[b]) * gmv[b];
                    float v3 = (__builtin_fmaf(oX[b][4 * g + 3], 1.0f / 2048.0f, oH[b][4 * g + 3]) + b2v[b]) * gmv[b];
                    quad_transpose(v0, v1, v2, v3, lj);
                    oH[b][4 * g + 0] = v0; oH[b][4 * g + 1] = v1; oH[b][4 * g + 2] = v2; oH[b][4 * g + 3] = v3;
                }
            }
            // (residual rows long here;) next tile's positions 0 and 1 landed -- ARES: and its A stages 0 / 1; stages 2 / 3, the youngest
            // loads, may still be on their way
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ARES ? 2 * S::NA : 0) : "memory");
            H2C_T(105);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = m0 + 32 * wm + 4 * lho + 8 * g + lj;       // (recomputed: the rows of the residual request, not kept in registers)
                const bool rokg = row < d.M;
                const int rbg = (rokg ? row : 0) * C;
#pragma unroll
                for (int b = 0; b < NB2; ++b) {
#ifdef H2C_EXP_NOEPI
                    if (rokg && oH[b][4 * g] == 123.456f) {
#else
                    if (rokg) {
#endif
                        f32x4 o = {oH[b][4 * g + 0], oH[b][4 * g + 1], oH[b][4 * g + 2], oH[b][4 * g + 3]};
                        o[0] += rv[g][b][0]; o[1] += rv[g][b][1]; o[2] += rv[g][b][2]; o[3] += rv[g][b][3];
                        *(f32x4*)(d.out + rbg + col_of(b) + (lio & ~3)) = o;
                    }
                }
            }
            H2C_T(106);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <int C_, int HID_, int HC_, int BM_ = 128, int NSUB_ = 1, int NBUF_ = 3, bool ARES_ = false>
int launch_h2c(const lvae_mlp_desc* d, hipStream_t st) {
    using S = H2C<C_, HID_, HC_, BM_, NSUB_, NBUF_, ARES_>;
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)mlp_h2c_kernel<C_, HID_, HC_, BM_, NSUB_, NBUF_, ARES_>, S::LDS)) return ae;
    const int n_cu = lvae_cu_count();                       // per device (gemm_common.h), like the LDS attribute above
    if ((long)d->M * S::C * 4 > 0x7fffffffL) return -22;    // 32-bit row offsets in the epilogue, one buffer descriptor per tile base
    const int n_tiles = (d->M + S::BM - 1) / S::BM;
    const int grid = n_tiles < n_cu ? n_tiles : n_cu;      // one persistent workgroup per CU (it owns the whole LDS)
    hipLaunchKernelGGL((mlp_h2c_kernel<C_, HID_, HC_, BM_, NSUB_, NBUF_, ARES_>), dim3(grid), dim3(512), S::LDS, st, *d, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// fc1 -> GELU -> fc2 -> residual of one block as one launch (include/lvae_hip.h); -22: no fused form of this block shape
extern "C" int lvae_mlp_h2f(const lvae_mlp_desc* d, void* stream) {
    if (!d || !d->y || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->gamma || !d->res || !d->out || d->M <= 0) return -22;
    if (d->C == 192 && d->hid == 384) return launch_h2c<192, 384, 128>(d, (hipStream_t)stream);
#ifdef H2C_EXP_NOARES
    if (d->C == 128 && d->hid == 192) return launch_h2c<128, 192, 64>(d, (hipStream_t)stream);
#else
    if (d->C == 128 && d->hid == 192) return launch_h2c<128, 192, 64, 128, 1, 3, true>(d, (hipStream_t)stream);
#endif
    if (d->C == 384 && d->hid == 768) return launch_h2c<384, 768, 128, 64, 3, 4>(d, (hipStream_t)stream);
    // (C = 256 / hidden = 448, 512 as <256, hid, 64, 128, 2, 4> was instantiated and measured: bit-identical, 30 spilled registers, 134.7 against
    //  124.3 us at M = 49152, 66.2 against 68.9 at 24576: profiles/r04_mlp_h2c_384x768.txt -- not built into the library)
    return -22;
}
