/*
 * oracle/rans_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked/imported by the product path).
 *
 * Plain-C restatement of the host entropy-coding primitives the reference reaches through the
 * un-vendored third-party package CompressAI (PyPI `compressai`, version UNPINNED by the reference:
 * /root/reference/README.md:67 names it without a version; setup.py lists no requirements):
 *
 *   - compressai/cpp_exts/ops/ops.cpp::pmf_to_quantized_cdf          
 *   - compressai/cpp_exts/rans/rans_interface.cpp + ryg rans64.h      (SURVEY.md N1,N2 / Appendix A.3)
 *       BufferedRansEncoder::encode_with_indexes + flush, RansDecoder::decode_with_indexes
 *
 * Reference call sites that fix the semantics (the reference has no tests / golden bitstreams):
 *   lvae/models/qarv/model.py:107 (compress), :113 (decompress), :124 (update -> pmf_to_quantized_cdf)
 *   lvae/models/qresvae/model.py:325,339,356
 *
 * PARITY UNPINNED w.r.t. real CompressAI bitstreams: CompressAI is absent from this image and from
 * /root/reference, and the reference ships no known-answer vectors.  What pins this file:
 *   (i)  round trips on random / adversarial streams incl. bypass escapes (tests/test_oracle_rans.py),
 *   (ii) CDF post-conditions (cdf[0]=0, last=65536, strictly increasing),
 *   (iii) coded size within ~1% of sum(-log2 P) from the eval-mode likelihood (qarv/model.py:95-96).
 *
 * Style: deliberately naive (symbol list pushed forward, popped backward, linear CDF search) so that it
 * reads like the published algorithm; the product coder (lossy-vae_amd/csrc/rans_host.cpp) is an
 * independent, optimised implementation checked against this one bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION 16
#define BYPASS_PRECISION 4
#define MAX_BYPASS_VAL ((1 << BYPASS_PRECISION) - 1)
#define RANS64_L (1ull << 31)

/* ---------------------------------------------------------------- pmf_to_quantized_cdf (A.2) */
/* returns 0 ok, -1 invalid pmf entry, -2 zero total, -3 could not steal */
int oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf /* n+1 */) {
    int i, j;
    for (i = 0; i < n; ++i)
        if (pmf[i] < 0 || !isfinite(pmf[i])) return -1;
    cdf[0] = 0;
    for (i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision));
    uint32_t total = 0;
    for (i = 0; i <= n; ++i) total += cdf[i];
    if (total == 0) return -2;
    for (i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
    for (i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
    cdf[n] = 1u << precision;
    for (i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (j = 0; j < n; ++j) {
                uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) { best_freq = freq; best_steal = j; }
            }
            if (best_steal == -1) return -3;
            if (best_steal < i) { for (j = best_steal + 1; j <= i; ++j) cdf[j]--; }
            else                { for (j = i + 1; j <= best_steal; ++j) cdf[j]++; }
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- rANS encode (A.3) */
typedef struct { uint16_t start; uint16_t range; int bypass; } sym_t;
typedef struct { sym_t *v; size_t n, cap; } symvec_t;

static int push(symvec_t *s, uint16_t start, uint16_t range, int bypass) {
    if (s->n == s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 1024;
        sym_t *nv = (sym_t *)realloc(s->v, nc * sizeof(sym_t));
        if (!nv) return -1;
        s->v = nv; s->cap = nc;
    }
    s->v[s->n].start = start; s->v[s->n].range = range; s->v[s->n].bypass = bypass; s->n++;
    return 0;
}

/* cdfs: row-major int32 [n_rows][row_stride].  Returns bytes written (multiple of 4) or <0. */
long oracle_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, size_t n,
                                     const int32_t *cdfs, int row_stride, const int32_t *cdf_sizes,
                                     const int32_t *offsets, uint8_t *out, size_t out_cap) {
    symvec_t sv = {0, 0, 0};
    size_t i;
    for (i = 0; i < n; ++i) {
        const int32_t cdf_idx = indexes[i];
        const int32_t *cdf = cdfs + (size_t)cdf_idx * row_stride;
        const int32_t max_value = cdf_sizes[cdf_idx] - 2;
        int32_t value = symbols[i] - offsets[cdf_idx];
        uint32_t raw_val = 0;
        if (value < 0) { raw_val = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw_val = (uint32_t)(2 * (value - max_value)); value = max_value; }
        if (push(&sv, (uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0)) goto oom;
        if (value == max_value) {
            int32_t n_bypass = 0;
            /* upstream shifts a uint32 by up to 32 here (UB for raw_val >= 2^28); bound it: |v| < 2^27 is the supported range */
            while (n_bypass < 8 && (raw_val >> (n_bypass * BYPASS_PRECISION)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= MAX_BYPASS_VAL) {
                if (push(&sv, MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, 1)) goto oom;
                val -= MAX_BYPASS_VAL;
            }
            if (push(&sv, (uint16_t)val, (uint16_t)(val + 1), 1)) goto oom;
            for (int32_t j = 0; j < n_bypass; ++j) {
                const int32_t v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL;
                if (push(&sv, (uint16_t)v, (uint16_t)(v + 1), 1)) goto oom;
            }
        }
    }
    {
        /* flush: pop back-to-front, write 32-bit words backwards */
        size_t nwords = sv.n + 2;
        uint32_t *buf = (uint32_t *)malloc(nwords * sizeof(uint32_t));
        if (!buf) goto oom;
        uint32_t *ptr = buf + nwords;
        uint64_t x = RANS64_L;
        while (sv.n > 0) {
            sym_t s = sv.v[--sv.n];
            if (!s.bypass) {
                uint64_t x_max = ((RANS64_L >> PRECISION) << 32) * s.range;
                if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
                x = ((x / s.range) << PRECISION) + (x % s.range) + s.start;
            } else {
                uint32_t freq = 1u << (16 - BYPASS_PRECISION);
                uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
                if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
                x = (x << BYPASS_PRECISION) | s.start;
            }
        }
        ptr -= 2;
        ptr[0] = (uint32_t)(x >> 0);
        ptr[1] = (uint32_t)(x >> 32);
        size_t nbytes = (size_t)((buf + nwords) - ptr) * sizeof(uint32_t);
        long ret;
        if (nbytes > out_cap) ret = -2;
        else { memcpy(out, ptr, nbytes); ret = (long)nbytes; }
        free(buf); free(sv.v);
        return ret;
    }
oom:
    free(sv.v);
    return -1;
}

/* ---------------------------------------------------------------- rANS decode (A.3) */
static uint32_t dec_get_bits(uint64_t *r, const uint32_t **pptr, uint32_t n_bits) {
    uint64_t x = *r;
    uint32_t val = (uint32_t)(x & ((1u << n_bits) - 1));
    x >>= n_bits;
    if (x < RANS64_L) { x = (x << 32) | **pptr; *pptr += 1; }
    *r = x;
    return val;
}

int oracle_rans_decode_with_indexes(const uint8_t *in, size_t in_len, const int32_t *indexes, size_t n,
                                    const int32_t *cdfs, int row_stride, const int32_t *cdf_sizes,
                                    const int32_t *offsets, int32_t *out) {
    if (in_len < 8 || (in_len & 3)) return -1;
    /* copy to an aligned, zero-padded word buffer (decoder may read one word past the end, as upstream) */
    size_t nwords = in_len / 4;
    uint32_t *w = (uint32_t *)calloc(nwords + 4, sizeof(uint32_t));
    if (!w) return -2;
    memcpy(w, in, in_len);
    const uint32_t *ptr = w;
    uint64_t x = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32);
    ptr += 2;
    for (size_t i = 0; i < n; ++i) {
        const int32_t cdf_idx = indexes[i];
        const int32_t *cdf = cdfs + (size_t)cdf_idx * row_stride;
        const int32_t size = cdf_sizes[cdf_idx];
        const int32_t max_value = size - 2;
        const int32_t offset = offsets[cdf_idx];
        const uint32_t cum_freq = (uint32_t)(x & ((1u << PRECISION) - 1));
        int32_t k = 0;
        while (k < size && !((uint32_t)cdf[k] > cum_freq)) ++k; /* std::find_if(first > cum_freq) */
        const int32_t s = k - 1;
        {
            uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
            x = (uint64_t)freq * (x >> PRECISION) + (x & ((1ull << PRECISION) - 1)) - start;
            if (x < RANS64_L) { x = (x << 32) | *ptr; ptr += 1; }
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(&x, &ptr, BYPASS_PRECISION);
            int32_t n_bypass = val;
            while (val == MAX_BYPASS_VAL) {
                val = (int32_t)dec_get_bits(&x, &ptr, BYPASS_PRECISION);
                n_bypass += val;
            }
            int32_t raw_val = 0;
            for (int32_t j = 0; j < n_bypass; ++j) {
                val = (int32_t)dec_get_bits(&x, &ptr, BYPASS_PRECISION);
                raw_val |= val << (j * BYPASS_PRECISION);
            }
            value = raw_val >> 1;
            if (raw_val & 1) value = -value - 1;
            else value += max_value;
        }
        out[i] = value + offset;
        if ((size_t)(ptr - w) > nwords + 2) { free(w); return -3; } /* ran off the stream */
    }
    free(w);
    return 0;
}
