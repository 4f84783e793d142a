/* A C99 client of the drop-in boundary (include/lvae_hip.h): what a non-Python integrator links.  Builds the Gaussian CDF rows,
 * codes 100 000 symbols drawn around the rows' supports (escapes included) with lvae_rans_encode_with_indexes, decodes them with
 * lvae_rans_decode_with_indexes and compares; then checks the error codes the header documents.  Host entry points only (no GPU).
 * Prints "ok <abi> <bytes>" and returns 0, or a message and 1.   tests/test_abi.py::test_c99_client_round_trips_the_coder */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lvae_hip.h"

static unsigned long long rng = 0x9E3779B97F4A7C15ull;
static unsigned next_u32(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 32); }

int main(void) {
    enum { NS = 64, STRIDE = 512, N = 100000 };
    static float table[NS];
    static int32_t qcdf[NS * STRIDE], cdf_len[NS], offset[NS], sym[N], back[N];
    static uint8_t idx[N], out[8 * N];
    int i;
    for (i = 0; i < NS; ++i) table[i] = (float)exp(log(0.11) + (log(20.0) - log(0.11)) * i / (NS - 1));
    const int maxlen = lvae_build_gaussian_tables(table, NS, 6.109410204869, 0, qcdf, STRIDE, cdf_len, offset);
    if (maxlen <= 0 || maxlen > STRIDE) { printf("lvae_build_gaussian_tables -> %d\n", maxlen); return 1; }
    for (i = 0; i < NS; ++i) {
        if (cdf_len[i] < 3 || cdf_len[i] > maxlen || qcdf[i * STRIDE] != 0 || qcdf[i * STRIDE + cdf_len[i] - 1] != 65536) { printf("row %d malformed\n", i); return 1; }
    }
    for (i = 0; i < N; ++i) {
        const int r = (int)(next_u32() % NS), half = (cdf_len[r] - 2) / 2;
        idx[i] = (uint8_t)r;
        sym[i] = (int)(next_u32() % (unsigned)(2 * half + 9)) - half - 4;          /* a few values beyond the row's support: bypass-coded */
    }
    const long nb = lvae_rans_encode_with_indexes(sym, idx, N, qcdf, STRIDE, cdf_len, offset, out, sizeof out);
    if (nb < 8 || (nb & 3)) { printf("encode -> %ld\n", nb); return 1; }
    int rc = lvae_rans_decode_with_indexes(out, (size_t)nb, idx, N, qcdf, STRIDE, cdf_len, offset, back);
    if (rc != 0 || memcmp(sym, back, sizeof sym) != 0) { printf("decode -> %d, symbols %s\n", rc, rc ? "n/a" : "differ"); return 1; }
    if (lvae_rans_encode_with_indexes(sym, idx, N, qcdf, STRIDE, cdf_len, offset, out, 16) != -2) { printf("short buffer not refused\n"); return 1; }
    {   /* index 255 in a 256-row view of the same tables whose rows 64 .. 255 are empty (length 0): a bad index, -4 */
        static int32_t len2[256], off2[256], q2[256 * STRIDE];
        memcpy(len2, cdf_len, sizeof(int32_t) * NS);
        memcpy(off2, offset, sizeof(int32_t) * NS);
        memcpy(q2, qcdf, sizeof qcdf);
        idx[7] = 255;
        if (lvae_rans_encode_with_indexes(sym, idx, N, q2, STRIDE, len2, off2, out, sizeof out) != -4) { printf("bad index not refused\n"); return 1; }
    }
    if (lvae_rans_decode_with_indexes(out, 4, idx, N, qcdf, STRIDE, cdf_len, offset, back) >= 0) { printf("truncated stream not refused\n"); return 1; }
    printf("ok %d %ld\n", lvae_abi_version(), nb);
    return 0;
}
