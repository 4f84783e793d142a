// rans_host.cpp -- host entropy coder of liblvae_hip.so (C ABI in include/lvae_hip.h).
//
// MI355X-native replacement for the CompressAI pybind11 entry points the reference reaches at
// lvae/models/qarv/model.py:107 (encode_with_indexes), :113 (decode_with_indexes), :124 (pmf_to_quantized_cdf)
// and lvae/models/qresvae/model.py:325,339,356.  The range coder stays on the host (serial by nature) but is
// fed directly by GPU-produced uint8 scale indexes / int32 symbols in pinned buffers: no Python lists, symbols
// walked back-to-front in place (no intermediate symbol vector), bucket-LUT decode, and N independent streams
// (images x latent blocks) coded on N host threads.
//
// Bit-compatible with oracle/rans_oracle.c (the plain restatement of CompressAI's coder); checked in
// tests/test_host_coder.py.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/lvae_hip.h"

namespace {
constexpr uint32_t kPrecision = 16;
constexpr uint32_t kBypassPrecision = 4;
constexpr int32_t kMaxBypassVal = (1 << kBypassPrecision) - 1;
constexpr uint64_t kRansL = 1ull << 31;
// Legal length of a CDF row (cdf_len = own symbols + escape symbol + 1): symbol ids are bytes in the decoder's tables, so a row has at
// most 256 intervals -- ONE limit for the encoder and the decoder (the encoder used to take 258 and write streams the decoder
// refused).  The encoder needs two intervals or more (a one-interval row has frequency 2^16, which its 16-bit entry cannot hold); the
// decoder also reads the one-interval row (every symbol an escape), which the published coder can write.
constexpr int32_t kMaxCdfLen = 257, kMinCdfLenEnc = 3, kMinCdfLenDec = 2;

struct BackWriter {
    uint32_t* base;
    uint32_t* ptr;
    bool overflow = false;
    inline void put(uint32_t w) {
        if (ptr == base) { overflow = true; return; }
        *--ptr = w;
    }
};

inline void enc_put(uint64_t& x, BackWriter& w, uint32_t start, uint32_t freq) {
    const uint64_t x_max = ((kRansL >> kPrecision) << 32) * freq;
    if (x >= x_max) { w.put((uint32_t)x); x >>= 32; }
    x = ((x / freq) << kPrecision) + (x % freq) + start;
}
inline void enc_put_bits(uint64_t& x, BackWriter& w, uint32_t val) {
    const uint32_t freq = 1u << (16 - kBypassPrecision);
    const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) { w.put((uint32_t)x); x >>= 32; }
    x = (x << kBypassPrecision) | val;
}
inline uint32_t dec_get_bits(uint64_t& x, const uint32_t*& ptr, const uint32_t* end, bool& overrun) {
    const uint32_t val = (uint32_t)(x & ((1u << kBypassPrecision) - 1));
    x >>= kBypassPrecision;
    if (x < kRansL) {
        uint32_t w = 0;
        if (ptr < end) w = *ptr; else overrun = true;
        ++ptr;
        x = (x << 32) | w;
    }
    return val;
}
}  // namespace

extern "C" int lvae_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf) {
    if (n <= 0 || precision <= 0 || precision > 16) return -22;
    for (int i = 0; i < n; ++i)
        if (pmf[i] < 0 || !std::isfinite(pmf[i])) return -1;
    cdf[0] = 0;
    const float scale = (float)(1 << precision);
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)std::round(pmf[i] * scale);
    uint32_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];
    if (total == 0) return -2;
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best_freq = ~0u;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best_freq) { best_freq = f; best = j; }
        }
        if (best < 0) return -3;
        if (best < i) for (int j = best + 1; j <= i; ++j) cdf[j]--;
        else          for (int j = i + 1; j <= best; ++j) cdf[j]++;
    }
    return 0;
}

extern "C" int lvae_build_gaussian_tables(const float* scale_table, int n_scales, double multiplier, int cdf_form,
                                          int32_t* qcdf, int row_stride, int32_t* cdf_len, int32_t* offset) {
    if (n_scales <= 0) return -22;
    int max_len = 0;
    std::vector<int> centers(n_scales);
    const float mult = (float)multiplier;   // torch: float32 tensor * python scalar -> float32 product
    for (int i = 0; i < n_scales; ++i) {
        centers[i] = (int)std::ceil(scale_table[i] * mult);
        max_len = std::max(max_len, 2 * centers[i] + 1);
    }
    if (max_len + 2 > row_stride) return -2;
    // erf / erfc are evaluated in double and rounded once to float: a correctly rounded fp32 erf whatever the C library's erff
    // does in its last ulp, so the tables (and with them every bitstream) do not depend on the libm / torch / device in use.
    // Bit-identical to the reference-built tables of tests/golden/{discretized_gaussian,gaussian_conditional}_tables.npz.
    auto phi = [cdf_form](float v) -> float {
        if (cdf_form == 0) return 0.5f * (1.0f + (float)erf((double)(v / (float)M_SQRT2)));
        return 0.5f * (float)erfc((double)(-(float)M_SQRT1_2 * v));
    };
    std::vector<float> pmf(max_len + 1);
    std::vector<uint32_t> cdf(max_len + 2);
    for (int i = 0; i < n_scales; ++i) {
        const int c = centers[i], len = 2 * c + 1;
        const float s = scale_table[i];
        float lower0 = 0.f;
        for (int k = 0; k < len; ++k) {
            const float a = (float)std::abs(k - c);
            const float up = phi((0.5f - a) / s), lo = phi((-0.5f - a) / s);
            pmf[k] = up - lo;
            if (k == 0) lower0 = lo;
        }
        pmf[len] = 2.0f * lower0;
        const int rc = lvae_pmf_to_quantized_cdf(pmf.data(), len + 1, 16, cdf.data());
        if (rc) return rc;
        int32_t* row = qcdf + (size_t)i * row_stride;
        std::memset(row, 0, sizeof(int32_t) * row_stride);
        for (int k = 0; k < len + 2; ++k) row[k] = (int32_t)cdf[k];
        cdf_len[i] = len + 2;
        offset[i] = -c;
    }
    return max_len + 2;
}

// Encoder symbol entries.  rANS's encode step is x' = ((x / freq) << 16) + (x % freq) + start: a 64-bit division (25-40 cycles of
// latency on the host cores) in the serial dependency chain of EVERY symbol -- 7.8 ns per symbol against the decoder's 2.4
// (tools/rans_bench.py), and the encoder's backlog is what the GPU's last latent block waits behind.  The quotient
// comes from a reciprocal instead (Alverson, "Integer division using reciprocals"; the form of ryg's rans64.h): with
// l = ceil(log2 freq) and m = ceil(2^(63 + l) / freq), floor(x / freq) == mulhi64(x, m) >> (l - 1) for every x < 2^63 (the coder's
// state is < 2^47 * freq <= 2^63 after the renormalisation check), and x' = x + start + q * (65536 - freq).  freq == 1 has no
// reciprocal below 1: m = 2^64 - 1, shift 0 gives x - 1 for x >= 1, compensated in the bias.  An entry is built the first time a
// stream meets its (row, value) -- two divisions, ~10 ns, a few hundred to a few thousand distinct pairs per stream -- in tables
// that live for ONE call: no shared state, the C ABI stays as it was.  Same bytes as the division form
// (tests/test_host_coder.py::test_reciprocal_encoder_equals_division_form, and every stream test against oracle/rans_oracle.c).
namespace {
struct EncEnt { uint64_t rcp; uint32_t bias; uint16_t freq; uint8_t shift; uint8_t ready; };      // 16 bytes
struct EncRow {
    int32_t offset = 0, max_value = 0;
    EncEnt e[kMaxCdfLen - 1];                            // values 0 .. max_value (the escape symbol is entry max_value)
};
inline void enc_ent_init(EncEnt& s, uint32_t start, uint32_t freq) {
    s.freq = (uint16_t)freq;                             // 1 <= freq <= 65535: a row has at least two symbols of frequency >= 1
    s.ready = 1;
    if (freq < 2) {
        s.rcp = ~0ull; s.shift = 0; s.bias = start + (1u << kPrecision) - 1;
    } else {
        uint32_t shift = 0;
        while (freq > (1u << shift)) ++shift;
        // ((uint128)1 << (shift + 63)) + freq - 1) / freq by long division in 64-bit words
        uint64_t x0 = freq - 1;
        const uint64_t x1 = 1ull << (shift + 31);
        const uint64_t t1 = x1 / freq;
        x0 += (x1 % freq) << 32;
        const uint64_t t0 = x0 / freq;
        s.rcp = t0 + (t1 << 32);
        s.shift = (uint8_t)(shift - 1);
        s.bias = start;
    }
}
inline void enc_put_ent(uint64_t& x, BackWriter& w, const EncEnt& s) {
    const uint64_t x_max = (uint64_t)s.freq << 47;       // ((kRansL >> kPrecision) << 32) * freq
    if (x >= x_max) { w.put((uint32_t)x); x >>= 32; }
    const uint64_t q = (uint64_t)(((unsigned __int128)x * s.rcp) >> 64) >> s.shift;
    x = x + s.bias + q * (uint64_t)((1u << kPrecision) - s.freq);
}
struct EncTabs {
    std::unique_ptr<EncRow> rows[256];
    // -> nullptr: the row's cdf length is out of range
    EncRow* row(int r, const int32_t* cdf_len, const int32_t* offset) {
        if (rows[r]) return rows[r].get();
        if (cdf_len[r] < kMinCdfLenEnc || cdf_len[r] > kMaxCdfLen) return nullptr;
        const int32_t mv = cdf_len[r] - 2;
        rows[r].reset(new (std::nothrow) EncRow);
        if (!rows[r]) return nullptr;
        rows[r]->offset = offset[r];
        rows[r]->max_value = mv;
        for (auto& e : rows[r]->e) e.ready = 0;
        return rows[r].get();
    }
};
}  // namespace

extern "C" long lvae_rans_encode_with_indexes(const int32_t* sym, const uint8_t* idx, size_t n,
                                              const int32_t* qcdf, int row_stride, const int32_t* cdf_len,
                                              const int32_t* offset, uint8_t* out, size_t out_cap) {
    if (out_cap < 8) return -2;
    // words are written backwards from the end of `out` (4-byte aligned view), then moved to the front
    uint8_t* aligned = (uint8_t*)(((uintptr_t)out + 3) & ~(uintptr_t)3);
    const size_t cap_words = (out_cap - (size_t)(aligned - out)) / 4;
    BackWriter w{(uint32_t*)aligned, (uint32_t*)aligned + cap_words};
    uint64_t x = kRansL;
    std::unique_ptr<EncTabs> T(new (std::nothrow) EncTabs);
    if (!T) return -12;
    EncRow* cur = nullptr;
    int32_t cur_row = -1;
    for (size_t ii = n; ii-- > 0;) {
        const int32_t row_i = idx[ii];
        if (row_i != cur_row) {
            cur = T->row(row_i, cdf_len, offset);
            if (!cur) return -4;
            cur_row = row_i;
        }
        const int32_t max_value = cur->max_value;
        const int32_t value = sym[ii] - cur->offset;
        const bool in_range = value >= 0 && value < max_value;
        EncEnt& e = cur->e[in_range ? value : max_value];
        if (__builtin_expect(!e.ready, 0)) {
            const int32_t* cdf = qcdf + (size_t)row_i * row_stride;
            const int32_t v = in_range ? value : max_value;
            enc_ent_init(e, (uint32_t)cdf[v], (uint32_t)(cdf[v + 1] - cdf[v]));
        }
        if (__builtin_expect(in_range, 1)) {
            enc_put_ent(x, w, e);
            continue;
        }
        uint32_t raw;
        if (value < 0) raw = (uint32_t)(-2 * (int64_t)value - 1);
        else           raw = (uint32_t)(2 * ((int64_t)value - max_value));
        int32_t n_bypass = 0;
        while (n_bypass < 8 && (raw >> (n_bypass * kBypassPrecision)) != 0) ++n_bypass;
        // forward order is: [escape symbol][count nibbles: 15,15,..,rem][raw nibbles LSB first]; emit reversed
        for (int32_t j = n_bypass - 1; j >= 0; --j) enc_put_bits(x, w, (raw >> (j * kBypassPrecision)) & kMaxBypassVal);
        const int32_t n15 = n_bypass / kMaxBypassVal, rem = n_bypass % kMaxBypassVal;
        enc_put_bits(x, w, (uint32_t)rem);
        for (int32_t j = 0; j < n15; ++j) enc_put_bits(x, w, (uint32_t)kMaxBypassVal);
        enc_put_ent(x, w, e);
    }
    w.put((uint32_t)(x >> 32));
    w.put((uint32_t)x);
    if (w.overflow) return -2;
    const size_t nbytes = (size_t)(((uint32_t*)aligned + cap_words) - w.ptr) * 4;
    std::memmove(out, w.ptr, nbytes);
    return (long)nbytes;
}

// the division form of the encode step (the formula as published; tests compare the reciprocal form above with it, state by state)
extern "C" int lvae_rans_enc_step_selftest(uint64_t x, uint32_t start, uint32_t freq, uint64_t* by_division, uint64_t* by_reciprocal) {
    if (freq < 1 || freq > 65535 || !by_division || !by_reciprocal) return -22;
    uint32_t sink[4];
    BackWriter w1{sink, sink + 4}, w2{sink, sink + 4};
    uint64_t a = x, b = x;
    enc_put(a, w1, start, freq);
    EncEnt e;
    enc_ent_init(e, start, freq);
    enc_put_ent(b, w2, e);
    *by_division = a; *by_reciprocal = b;
    return (w1.ptr == w2.ptr) ? 0 : 1;
}

// Per-row decode tables, built lazily for the rows a stream touches (1.3 KB per row):
//   lut[row][b]   = largest s with cdf[s] <= (b << 8)                     (256 buckets of 256 counts)
//   freq[row][b], start[row][b] = freq(s), cdf[s]  if the whole bucket lies inside symbol s ("pure"), else freq = 0
//   info[row]     = the row's MOST PROBABLE symbol (start, freq, id) -- the escape symbol excluded -- and its escape id
// The serial dependency of a rANS stream is  state -> cf -> symbol -> (start, freq) -> state.  Upstream's linear find_if, and
// the first form here (bucket -> forward scan of data-dependent length -> two cdf loads), put dependent loads and a poorly
// predictable branch into that chain for every symbol.  Three forms, fastest first:
//   1. most-probable-symbol path (round 5): start and freq of the row's mode depend on the scale index alone, i.e. they are known
//      BEFORE the state is; if cf falls inside the mode -- a compare the branch predictor learns on compressible data: by definition
//      the mode is what a stream mostly holds -- the state update is  shift -> multiply -> add  with no table load in the chain
//      (~5 cycles instead of ~11).  A stream whose symbols are not mostly modes (hit rate below ~80 % over 1024 symbols: each miss is a
//      mispredicted branch) turns the path off for a while and probes again later -- except on rows whose mode alone holds >= 80 % of the
//      mass, where the model itself predicts the hit; the decoded symbols never depend on any of that.
//   2. pure bucket: most of a row's probability mass sits in symbols much wider than a bucket, so ONE pair of loads (freq, start:
//      two 16-bit tables, no unpacking shift in the chain) yields the update and the symbol id comes off the critical path;
//   3. only buckets that contain a boundary take the scan.
// Measured: tools/rans_bench.py, profiles/r05_rans_mps_path.txt.
// A table set belongs to ONE (qcdf, cdf_len) pair whose contents do not change while it lives.  It may be shared: by the streams of a
// batch call (decoded concurrently: a row is built by whoever needs it first, the others wait the fraction of a microsecond that takes)
// and by the nine per-block calls of a decode (lvae_decode_blocks) -- building the ~64 rows costs 25-35 us, which used to sit on the
// decode chain once per latent block and stream.
struct RowTab { uint16_t freq[256]; uint16_t start[256]; uint8_t lut[256]; };
struct RowInfo { uint32_t mfreq, mstart; int32_t msym, max_value; uint32_t mpeak; };       // mfreq = 0: the row has no most-probable-symbol path; mpeak = mfreq if the mode alone holds >= 80 % of the row's mass, else 0
struct LvaeDecTabs {
    RowTab tabs[256];
    RowInfo info[256];
    std::atomic<uint8_t> state[256];          // 0 = empty, 1 = being built, 2 = ready, 3 = invalid cdf length
    LvaeDecTabs() { for (auto& st : state) st.store(0, std::memory_order_relaxed); }
};

namespace {
inline void spin_pause() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
// A row whose mode holds at least this share of the mass (0.8 * 2^16) tries the mode path even while the stream-level switch is off: the
// model itself says the compare will mostly hit there (break-even of the path is a hit rate of ~0.77)
constexpr uint32_t kMpsPeak = 52429;
// -> false: the row's cdf length is out of range
inline bool ensure_row(LvaeDecTabs& D, int row, const int32_t* cdf, int32_t size) {
    uint8_t st = D.state[row].load(std::memory_order_acquire);
    if (st == 2) return true;
    for (;;) {
        if (st == 3) return false;
        if (st == 0 && D.state[row].compare_exchange_strong(st, 1, std::memory_order_acq_rel)) break;
        if (st == 2) return true;
        spin_pause();
        st = D.state[row].load(std::memory_order_acquire);
    }
    if (size < kMinCdfLenDec || size > kMaxCdfLen) { D.state[row].store(3, std::memory_order_release); return false; }
    RowTab& T = D.tabs[row];
    int32_t sidx = 0;
    for (int b = 0; b < 256; ++b) {
        const uint32_t v = (uint32_t)b << 8;
        while (sidx + 1 < size - 1 && (uint32_t)cdf[sidx + 1] <= v) ++sidx;
        T.lut[b] = (uint8_t)sidx;
        const uint32_t f = (uint32_t)(cdf[sidx + 1] - cdf[sidx]);
        // pure: the next boundary is beyond the bucket (f >= 256 then; f <= 65535 for a row of two symbols or more -- a one-symbol
        // row is all escape symbol and takes the scan)
        const bool pure = (uint32_t)cdf[sidx + 1] >= v + 256 && f <= 0xFFFFu;
        T.freq[b] = pure ? (uint16_t)f : (uint16_t)0;
        T.start[b] = pure ? (uint16_t)cdf[sidx] : (uint16_t)0;
    }
    // the most probable symbol among the table's own symbols 0 .. size - 3 (symbol size - 2 = max_value is the escape)
    RowInfo& R = D.info[row];
    R.max_value = size - 2;
    R.mfreq = 0; R.mstart = 0; R.msym = 0;
    for (int32_t s = 0; s < size - 2; ++s) {
        const uint32_t f = (uint32_t)(cdf[s + 1] - cdf[s]);
        if (cdf[s] >= 0 && cdf[s + 1] <= 65536 && cdf[s + 1] > cdf[s] && f > R.mfreq) { R.mfreq = f; R.mstart = (uint32_t)cdf[s]; R.msym = s; }
    }
    R.mpeak = R.mfreq >= kMpsPeak ? R.mfreq : 0;
    D.state[row].store(2, std::memory_order_release);
    return true;
}

constexpr size_t kMpsWindow = 1024;           // symbols between two looks at the most-probable-symbol path's hit rate
constexpr uint32_t kMpsMaxMiss = 200;         // misses per window above which the path is switched off (a miss = a mispredicted branch)
constexpr int kMpsCoolWindows = 8;            // ... for this many windows, then probed again; doubled (up to kMpsCoolMax) every time a probe fails
constexpr int kMpsCoolMax = 512;

int decode_stream(const uint8_t* in, size_t in_len, const uint8_t* idx, size_t n, const int32_t* qcdf, int row_stride,
                  const int32_t* cdf_len, const int32_t* offset, int32_t* sym_out, LvaeDecTabs& D) {
    if (in_len < 8 || (in_len & 3)) return -1;
    std::vector<uint32_t> tmp;
    const uint32_t* words;
    if (((uintptr_t)in & 3) == 0) words = (const uint32_t*)in;
    else { tmp.resize(in_len / 4); std::memcpy(tmp.data(), in, in_len); words = tmp.data(); }
    const uint32_t* ptr = words;
    const uint32_t* end = words + in_len / 4;
    uint64_t x = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32);
    ptr += 2;
    bool overrun = false;
    bool mine[256] = {false};                 // rows this stream has already seen ready (skips the atomic load in the symbol loop)
    bool use_mps = true;
    uint32_t miss = 0;
    int cool = 0, cool_len = kMpsCoolWindows;
    size_t next_look = kMpsWindow;
    for (size_t i = 0; i < n; ++i) {
        const int32_t row_i = idx[i];
        if (!mine[row_i]) {
            if (!ensure_row(D, row_i, qcdf + (size_t)row_i * row_stride, cdf_len[row_i])) return -4;
            mine[row_i] = true;
        }
        if (i == next_look) {
            if (use_mps) {
                if (miss > kMpsMaxMiss) { use_mps = false; cool = cool_len; cool_len = cool_len * 2 < kMpsCoolMax ? cool_len * 2 : kMpsCoolMax; }
                else cool_len = kMpsCoolWindows;
            } else if (--cool <= 0) use_mps = true;
            miss = 0;
            next_look += kMpsWindow;
        }
        const RowInfo& R = D.info[row_i];
        const uint32_t cf = (uint32_t)(x & 0xFFFF);
        const uint32_t dm = cf - R.mstart;
        if (__builtin_expect(dm < (use_mps ? R.mfreq : R.mpeak), 1)) {
            // the row's mode: start / freq came from the scale index, nothing in the chain but shift, multiply, add
            x = (uint64_t)R.mfreq * (x >> kPrecision) + dm;
            if (x < kRansL) {
                uint32_t wv = 0;
                if (ptr < end) wv = *ptr; else overrun = true;
                ++ptr;
                x = (x << 32) | wv;
                if (overrun) return -3;
            }
            sym_out[i] = R.msym + offset[row_i];
            continue;
        }
        ++miss;
        const int32_t* cdf = qcdf + (size_t)row_i * row_stride;
        const RowTab& T = D.tabs[row_i];
        uint32_t freq = T.freq[cf >> 8], start = T.start[cf >> 8];
        int32_t s = T.lut[cf >> 8];
        if (__builtin_expect(freq == 0, 0)) {
            while ((uint32_t)cdf[s + 1] <= cf) ++s;      // cdf[size-1] = 65536 > cf terminates the scan
            start = (uint32_t)cdf[s]; freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        }
        x = (uint64_t)freq * (x >> kPrecision) + cf - start;
        if (x < kRansL) {
            uint32_t wv = 0;
            if (ptr < end) wv = *ptr; else overrun = true;
            ++ptr;
            x = (x << 32) | wv;
        }
        int32_t value = s;
        if (value == R.max_value) {
            int32_t val = (int32_t)dec_get_bits(x, ptr, end, overrun);
            int32_t n_bypass = val;
            while (val == kMaxBypassVal) {
                val = (int32_t)dec_get_bits(x, ptr, end, overrun);
                n_bypass += val;
                if (overrun) return -3;
            }
            uint32_t raw = 0;
            for (int32_t j = 0; j < n_bypass; ++j) {
                val = (int32_t)dec_get_bits(x, ptr, end, overrun);
                if (j < 8) raw |= (uint32_t)val << (j * kBypassPrecision);
            }
            value = (int32_t)(raw >> 1);
            if (raw & 1) value = -value - 1;
            else value += R.max_value;
        }
        sym_out[i] = value + offset[row_i];
        if (overrun) return -3;
    }
    return 0;
}
}  // namespace

extern "C" int lvae_rans_decode_with_indexes(const uint8_t* in, size_t in_len, const uint8_t* idx, size_t n,
                                             const int32_t* qcdf, int row_stride, const int32_t* cdf_len,
                                             const int32_t* offset, int32_t* sym_out) {
    std::unique_ptr<LvaeDecTabs> D(new (std::nothrow) LvaeDecTabs);
    if (!D) return -12;
    try { return decode_stream(in, in_len, idx, n, qcdf, row_stride, cdf_len, offset, sym_out, *D); } catch (...) { return -12; }
}

namespace {
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

// Persistent worker pool.  The coder is called 2 x (1 + 9) times per batch by two pipeline-group threads; spawning up to 35
// std::threads per call put ~0.5-1 ms of thread creation on each call and made step times jittery.  Jobs are index ranges
// [0, n) claimed with an atomic counter; callers take part in their own job and several callers may have jobs in flight.
struct Job {
    int n = 0, max_workers = 0;
    std::atomic<int> next{0}, done{0}, workers{0};
    void (*run)(void*, int) = nullptr;
    void* ctx = nullptr;
    std::mutex m;
    std::condition_variable cv;
};

class Pool {
public:
    static Pool& get() { static Pool* p = new Pool;  return *p; }      // leaked on purpose: destroying a condition_variable with waiting
                                                                        // (detached) workers at process exit blocks in pthread_cond_destroy
    void submit(const std::shared_ptr<Job>& j) {
        { std::lock_guard<std::mutex> l(m_); q_.push_back(j); }
        epoch_.fetch_add(1, std::memory_order_release);           // the spinning workers see this without a futex round trip
        const int need = j->max_workers < (int)th_.size() ? j->max_workers : (int)th_.size();
        for (int i = 0; i < need; ++i) cv_.notify_one();          // (notify_all woke all 64 workers for a 4-stream job)
    }
    int size() const { return (int)th_.size(); }
private:
    Pool() {
        int n = (int)std::thread::hardware_concurrency();
        if (n < 2) n = 2;
        if (n > 64) n = 64;
        for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i < kSpinners); });
        for (auto& t : th_) t.detach();            // workers live for the process; they only touch heap-owned jobs
    }
    // The decoder calls the pool once per latent block (9 dependent calls per image, 0.1-1 ms of work each, a GPU segment in
    // between): a worker that goes to sleep on the condition variable after every call costs a futex wake-up (tens of microseconds) on
    // the critical path of every block.  The first kSpinners workers therefore poll the submit counter for a while (~0.5 ms: longer
    // than a GPU segment between two blocks) before they block; the others sleep at once.
    // (Keeping them awake for the whole of a decode loop -- no give-up while a loop is in flight -- was measured in round 5: the blocks'
    //  coder times did not move and the bench lost 1.4 %, profiles/r05_pool_hot_not_taken.txt: not taken.)
    static constexpr int kSpinners = 8;
    void loop(bool spinner) {
        for (;;) {
            std::shared_ptr<Job> j;
            if (spinner) {
                const unsigned seen = epoch_.load(std::memory_order_acquire);
                for (int it = 0; it < 20000 && epoch_.load(std::memory_order_acquire) == seen; ++it) cpu_relax();
            }
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return !q_.empty(); });
                // retire exhausted / saturated jobs at the front and take the first one that still wants a worker -- in ONE pass under the
                // lock: going back to the spin after retiring a job (as this loop once did) left a second pipeline group's job, already
                // queued behind it, without helpers for a whole spin period (~0.5 ms on the per-block calls of the decode chain)
                while (!q_.empty()) {
                    const std::shared_ptr<Job>& f = q_.front();
                    if (f->next.load() >= f->n || f->workers.load() >= f->max_workers) { q_.pop_front(); continue; }
                    j = f;
                    j->workers.fetch_add(1);
                    break;
                }
            }
            if (j) work(*j);
        }
    }
public:
    static void work(Job& j) {
        int did = 0;
        for (int i; (i = j.next.fetch_add(1)) < j.n;) { j.run(j.ctx, i); ++did; }
        if (did && j.done.fetch_add(did) + did == j.n) { std::lock_guard<std::mutex> l(j.m); j.cv.notify_all(); }
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned> epoch_{0};
    std::deque<std::shared_ptr<Job>> q_;
    std::vector<std::thread> th_;
};

template <class F>
void parallel_for(int n, int n_threads, F&& f) {
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n;
    if (n_threads <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
    auto j = std::make_shared<Job>();
    j->n = n;
    j->max_workers = n_threads - 1;                 // the caller is the n_threads-th worker
    j->ctx = (void*)&f;
    j->run = [](void* c, int i) { (*(typename std::remove_reference<F>::type*)c)(i); };
    Pool::get().submit(j);
    Pool::work(*j);
    for (int it = 0; it < 20000 && j->done.load(std::memory_order_acquire) < n; ++it) cpu_relax();   // the stragglers are usually microseconds away
    std::unique_lock<std::mutex> l(j->m);
    j->cv.wait(l, [&] { return j->done.load() >= n; });
}
}  // namespace

extern "C" int lvae_rans_encode_batch(int n_streams, const int32_t* const* sym, const uint8_t* const* idx,
                                      const size_t* n, const int32_t* qcdf, int row_stride, const int32_t* cdf_len,
                                      const int32_t* offset, uint8_t* const* out, const size_t* out_cap,
                                      long* out_len, int n_threads) {
    if (n_streams < 0) return -22;
    try {
        parallel_for(n_streams, n_threads, [&](int s) {
            out_len[s] = lvae_rans_encode_with_indexes(sym[s], idx[s], n[s], qcdf, row_stride, cdf_len, offset, out[s],
                                                       out_cap[s]);
        });
    } catch (...) { return -12; }       // (allocation of the job object; the C ABI lets no exception out)
    int rc = 0;
    for (int s = 0; s < n_streams; ++s) if (out_len[s] < 0) rc = (int)out_len[s];
    return rc;
}

// Non-blocking form of lvae_rans_encode_batch for lvae_encode_blocks (plan_runtime.cpp): `begin` hands the block's streams to the pool
// and returns at once -- the caller goes on to wait for the NEXT latent block's event, so a block whose coding takes longer than the
// GPU needs for the following block (the stride-16 blocks: 147 456 symbols per image) no longer delays the blocks behind it -- `end`
// takes part in whatever is left, waits, and returns the batch's status.  The pointer arrays are copied; the buffers they point to
// must stay valid until `end`.
struct LvaeEncJob {
    std::vector<const int32_t*> sym;
    std::vector<const uint8_t*> idx;
    std::vector<size_t> n, out_cap;
    std::vector<uint8_t*> out;
    long* out_len = nullptr;
    const int32_t *qcdf = nullptr, *cdf_len = nullptr, *offset = nullptr;
    int row_stride = 0;
    std::shared_ptr<Job> job;
    void run(int s) {
        out_len[s] = lvae_rans_encode_with_indexes(sym[s], idx[s], n[s], qcdf, row_stride, cdf_len, offset, out[s], out_cap[s]);
    }
};
LvaeEncJob* lvae_rans_encode_batch_begin(int n_streams, const int32_t* const* sym, const uint8_t* const* idx, const size_t* n,
                                         const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                                         uint8_t* const* out, const size_t* out_cap, long* out_len, int n_threads) {
    if (n_streams < 0) return nullptr;
    LvaeEncJob* e = new (std::nothrow) LvaeEncJob;
    if (!e) return nullptr;
    try {
    e->sym.assign(sym, sym + n_streams); e->idx.assign(idx, idx + n_streams); e->n.assign(n, n + n_streams);
    e->out.assign(out, out + n_streams); e->out_cap.assign(out_cap, out_cap + n_streams);
    e->out_len = out_len; e->qcdf = qcdf; e->cdf_len = cdf_len; e->offset = offset; e->row_stride = row_stride;
    for (int s = 0; s < n_streams; ++s) out_len[s] = 0;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 1) {                                 // one thread: coded here and now, like lvae_rans_encode_batch
        for (int s = 0; s < n_streams; ++s) e->run(s);
    } else if (n_streams > 0) {
        auto j = std::make_shared<Job>();
        j->n = n_streams;
        j->max_workers = n_threads < n_streams ? n_threads : n_streams;
        j->ctx = (void*)e;
        j->run = [](void* c, int i) { ((LvaeEncJob*)c)->run(i); };
        e->job = j;
        Pool::get().submit(j);
    }
    } catch (...) {                     // std::bad_alloc from the vectors / the job / the pool's queue: nothing was handed to the pool
        delete e;
        return nullptr;
    }
    return e;
}
int lvae_rans_encode_batch_end(LvaeEncJob* e) {
    if (!e) return -12;
    const int n = (int)e->sym.size();
    if (e->job) {
        Job& j = *e->job;
        Pool::work(j);
        for (int it = 0; it < 20000 && j.done.load(std::memory_order_acquire) < n; ++it) cpu_relax();
        std::unique_lock<std::mutex> l(j.m);
        j.cv.wait(l, [&] { return j.done.load() >= n; });
    }
    int rc = 0;
    for (int s = 0; s < n; ++s) if (e->out_len[s] < 0) rc = (int)e->out_len[s];
    delete e;
    return rc;
}

// lvae_rans_decode_batch with caller-owned decode tables (plan_runtime.cpp: one table set for the nine per-block calls of a decode);
// the tables must have been made for this (qcdf, cdf_len) by lvae_dec_tabs_new and die with lvae_dec_tabs_free
LvaeDecTabs* lvae_dec_tabs_new() { return new (std::nothrow) LvaeDecTabs; }
void lvae_dec_tabs_free(LvaeDecTabs* t) { delete t; }
int lvae_rans_decode_batch_tabs(int n_streams, const uint8_t* const* in, const size_t* in_len, const uint8_t* const* idx,
                                const size_t* n, const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                                int32_t* const* sym_out, int* status, int n_threads, LvaeDecTabs* tabs) {
    if (n_streams < 0 || !tabs) return -22;
    try {
        parallel_for(n_streams, n_threads, [&](int s) {
            int rc = -12;
            try { rc = decode_stream(in[s], in_len[s], idx[s], n[s], qcdf, row_stride, cdf_len, offset, sym_out[s], *tabs); } catch (...) {}
            status[s] = rc;
        });
    } catch (...) { return -12; }
    int rc = 0;
    for (int s = 0; s < n_streams; ++s) if (status[s] < 0) rc = status[s];
    return rc;
}

extern "C" int lvae_rans_decode_batch(int n_streams, const uint8_t* const* in, const size_t* in_len,
                                      const uint8_t* const* idx, const size_t* n, const int32_t* qcdf,
                                      int row_stride, const int32_t* cdf_len, const int32_t* offset,
                                      int32_t* const* sym_out, int* status, int n_threads) {
    if (n_streams < 0) return -22;
    std::unique_ptr<LvaeDecTabs> D(new (std::nothrow) LvaeDecTabs);      // shared by the streams of this call
    if (!D) return -12;
    return lvae_rans_decode_batch_tabs(n_streams, in, in_len, idx, n, qcdf, row_stride, cdf_len, offset, sym_out, status, n_threads, D.get());
}
