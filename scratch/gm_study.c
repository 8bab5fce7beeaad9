/* Round 5 design study (CPU): a MATCH model for the bases of the block format's frozen mode.
   Instead of a 2^24-row context table (one random memory access per base), a chain follows a pointer into the bases of
   EARLIER generations of the same call: an index of sampled K-mers (earliest occurrence wins) gives the pointer, the predicted
   base is read sequentially from the history, and the confidence follows the length of the verified match.
   Cost = sum of -log2 p over all bases (the coder's overhead is not in it).
   usage: gm_study file.fq [K] [table_bits] [sample_shift]                                                        */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
static int code_of(u8 c) { switch (c | 0x20) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': return 3; } return 0; }
#define MCAP 32
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    u8* fq = malloc(n); if (fread(fq, 1, n, f) != n) return 1; fclose(f);
    const int K = argc > 2 ? atoi(argv[2]) : 20, tb = argc > 3 ? atoi(argv[3]) : 23, ss = argc > 4 ? atoi(argv[4]) : 3;
    const int drop_m = argc > 5 ? atoi(argv[5]) : 8; const int use_rc = argc > 6 ? atoi(argv[6]) : 0; const int D = argc > 7 ? atoi(argv[7]) : 0;
    /* stage: base lines + '\n' */
    size_t nrec = 0; for (size_t i = 0; i < n; i++) nrec += fq[i] == '\n'; nrec /= 4;
    u8* st = malloc(n / 2 + nrec + 16); u64* soff = malloc((nrec + 1) * 8); u32* slen = malloc(nrec * 4);
    size_t p = 0, sp = 0;
    for (size_t r = 0; r < nrec; r++) {
        while (fq[p] != '\n') p++; p++;
        soff[r] = sp; size_t q = p; while (fq[q] != '\n') q++;
        slen[r] = (u32)(q - p); for (size_t i = p; i < q; i++) st[sp++] = (u8)code_of(fq[i]); st[sp++] = 0xFF;
        p = q + 1; while (fq[p] != '\n') p++; p++; while (fq[p] != '\n') p++; p++;
    }
    soff[nrec] = sp;
    const size_t br = 1024, nblocks = (nrec + br - 1) / br;
    size_t bound[48], ngen = 0; bound[0] = 0; { size_t b = (nblocks + 63) / 64; if (!b) b = 1; while (b < nblocks && ngen + 2 < 40) { bound[++ngen] = b; b = b * 2 > b + 1 ? b * 2 : b + 1; } bound[++ngen] = nblocks; }
    const u64 kmask = K < 32 ? ((1ull << (2 * K)) - 1) : ~0ull;
    u64* T = malloc(8ull << tb); memset(T, 0xFF, 8ull << tb);
    /* confidence: P(match) by m; F_other out of 4096 */
    double conf[MCAP + 1];
    for (int m = 0; m <= MCAP; m++) conf[m] = m < 4 ? 0.90 : m < 8 ? 0.95 : m < 16 ? 0.975 : m < 24 ? 0.985 : 0.992;
    double hits[MCAP + 1] = {0}, miss[MCAP + 1] = {0};
    double total_bits = 0; u64 total_bases = 0;
    u64 n_lookup = 0, n_found = 0, n_bad = 0, n_matched = 0, n_flat = 0, n_end = 0, n_drop = 0;
    for (size_t g = 0; g < ngen; g++) {
        const size_t r0 = bound[g] * br < nrec ? bound[g] * br : nrec, r1 = bound[g + 1] * br < nrec ? bound[g + 1] * br : nrec;
        const u64 limit = soff[r0];
        double gbits = 0; u64 gbases = 0;
        for (size_t r = r0; r < r1; r++) {
            u64 kmer = 0, rck = 0; u64 ptr = 0; int have = 0, m = 0, dir = 1; int pend = -1; u64 pend_ptr = 0; int pend_dir = 1;
            const u8* b = st + soff[r];
            for (u32 i = 0; i < slen[r]; i++) {
                const int c = b[i];
                double bits;
                if (pend >= 0 && (int)i >= pend) { have = 1; m = K; ptr = pend_ptr; dir = pend_dir; pend = -1; if (st[ptr] == 0xFF) have = 0; }
                if (have && st[ptr] == 0xFF) { have = 0; n_end++; }
                if (have) {
                    const int e = dir > 0 ? st[ptr] : 3 - st[ptr];
                    const double P = conf[m];
                    if (c == e) { bits = -log2(P); hits[m]++; m = m < MCAP ? m + 1 : MCAP; ptr += dir; }
                    else { bits = -log2((1 - P) / 3); miss[m]++; if (m < drop_m) { have = 0; n_drop++; } else { m = 0; ptr += dir; } }
                    if (have && dir < 0 && ptr == (u64)-1) have = 0;
                    n_matched++;
                } else { bits = 2; n_flat++; }
                gbits += bits; gbases++;
                kmer = ((kmer << 2) | (u64)c) & kmask;
                rck = (rck >> 2) | ((u64)(3 - c) << (2 * (K - 1)));
                if (use_rc && !have && pend < 0 && i + 1 >= (u32)K && i + 1 < slen[r]) {
                    const u64 h = rck * 0x9E3779B97F4A7C15ull;
                    if ((h >> (64 - ss)) == 0) {
                        n_lookup++;
                        const u64 e = T[(h << ss) >> (64 - tb)];
                        if (e != ~0ull && (e >> 24) < limit && (e >> 24) >= (u64)K + 1) {
                            if ((e & 0xFFFFFF) == ((h >> 8) & 0xFFFFFF)) { if (pend < 0) { pend = i + 1 + D; pend_ptr = (e >> 24) - K - 1 - D; pend_dir = -1; n_found++; if (pend_ptr > (1ull << 60)) pend = -1; } }
                            else n_bad++;
                        }
                    }
                }
                if (!have && pend < 0 && i + 1 >= (u32)K && i + 1 < slen[r]) {
                    const u64 h = kmer * 0x9E3779B97F4A7C15ull;
                    if ((h >> (64 - ss)) == 0) {
                        n_lookup++;
                        const u64 e = T[(h << ss) >> (64 - tb)];
                        if (e != ~0ull && (e >> 24) < limit) {
                            if ((e & 0xFFFFFF) == ((h >> 8) & 0xFFFFFF)) { if (pend < 0) { pend = i + 1 + D; pend_ptr = (e >> 24) + D; pend_dir = 1; n_found++; } }
                            else n_bad++;
                        }
                    }
                }
            }
        }
        total_bits += gbits; total_bases += gbases;
        printf("gen %zu: %.4f bit/base\n", g, gbits / (gbases ? gbases : 1));
        /* insert this generation (every stride-th record, as the counting passes) */
        if (g + 1 < ngen) {
            const size_t cnt = (bound[g + 1] - bound[g]) * br; size_t stride = (cnt + 524287) / 524288; if (!stride) stride = 1;
            for (size_t r = r0; r < r1; r += stride) {
                u64 kmer = 0; const u8* b = st + soff[r];
                for (u32 i = 0; i + 1 < slen[r]; i++) {
                    kmer = ((kmer << 2) | (u64)b[i]) & kmask;
                    if (i + 1 >= (u32)K) {
                        const u64 h = kmer * 0x9E3779B97F4A7C15ull;
                        if ((h >> (64 - ss)) == 0) {
                            const u64 v = ((soff[r] + i + 1) << 24) | ((h >> 8) & 0xFFFFFF);
                            u64* s = &T[(h << ss) >> (64 - tb)];
                            if (v < *s) *s = v;
                        }
                    }
                }
            }
        }
    }
    u64 used = 0; for (u64 i = 0; i < (1ull << tb); i++) used += T[i] != ~0ull;
    printf("K %d table 2^%d (%.1f %% used) sample 1/%d drop_m %d: %.0f B = %.4f bit/base; lookups %.2f/read found %.2f/read wrong-check %.3f/read; matched %.1f %% of bases; ends %.2f/read drops %.2f/read\n",
           K, tb, 100.0 * used / (double)(1ull << tb), 1 << ss, drop_m, total_bits / 8, total_bits / total_bases, (double)n_lookup / nrec, (double)n_found / nrec, (double)n_bad / nrec,
           100.0 * n_matched / total_bases, (double)n_end / nrec, (double)n_drop / nrec);
    printf("P(match) by m: "); for (int m = 0; m <= MCAP; m++) if (hits[m] + miss[m] > 0) printf("%d:%.4f(%.0f) ", m, hits[m] / (hits[m] + miss[m]), hits[m] + miss[m]); printf("\n");
    return 0;
}
