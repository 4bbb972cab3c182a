/*
 * stream_battery.c - statistical battery for the engine's random stream ON ITS PRODUCTION COUNTER LAYOUT.
 *
 * THIS IS TEST INFRASTRUCTURE (tests/test_stream_quality.py, tools/stream_battery_report.py), not product code.
 *
 * The stream is Philox4x32-R, R = 7 since stream revision 2 (Salmon et al., SC11: the smallest round count that passes
 * BigCrush; Random123's default 10 adds a margin), addressed by
 *     words = philox(key = env key (64 bit), ctr = (index, stage, keyword, tick))
 * A counter-based generator is only as good as its worst pair of NEIGHBOURING counters, and the kernels consume exactly
 * such neighbours together: the four words of a call are four consecutive auctions of a keyword (or the volumes of four
 * consecutive keywords); calls index j and j + 1 are the next four auctions; keyword k and k + 1 sit in adjacent lanes; tick t
 * and t + 1 are consecutive days; env e and e + 1 differ only in the key.  For each of these axes the battery draws the
 * words as the kernels address them and tests, per axis:
 *   - serial correlation at lags 1..4 of the word sequence along the axis (z-scores, N(0,1) under independence);
 *   - a 2-D chi-square of (word, neighbour along the axis) on their TOP bytes and on their BOTTOM bytes (256 x 256 cells),
 *     reported as z = (chi2 - dof) / sqrt(2 dof);
 *   - a chi-square of the top byte alone (uniformity) and the mean of the words' population counts.
 * and, for the words of one call, the same statistics between word positions (0,1), (1,2), (2,3), (0,3).
 * The DERIVED pair the IMPLICIT path takes from one auction word - the click and the competitor's 24-bit uniform - is
 * tested for independence on Philox words (the law tests of tests/test_oracle_scalar.py feed numpy PCG words: they check
 * the transforms, not the generator).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BAT_API __attribute__((visibility("default")))

static inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int rounds, uint32_t out[4])
{
    for (int r = 0; r < rounds; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* the product's env key: adcraft_amd/csrc/parts/kernels_misc.inc k_reset (splitmix64 of the seed and the global env id) */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint64_t env_key(uint64_t seed, uint64_t env) { return splitmix64(seed ^ splitmix64(env + 0x5851F42D4C957F2Dull)); }

enum { AX_WORDS = 0, AX_INDEX = 1, AX_KEYWORD = 2, AX_TICK = 3, AX_ENV = 4, AX_VOLUME = 5, AX_COUNT = 6 };
enum { STAT_PER_AXIS = 8 };      /* corr lag 1..4 | chi2 top bytes | chi2 bottom bytes | uniformity of the top byte | popcount mean */

typedef struct {
    double sx, sxx;            /* of the centred uniform u = (w + 0.5) / 2^32 - 0.5 */
    double sxy[4];             /* lag products */
    double n, npair[4];
    double pop;
} Moments;

/* one sequence of `len` words along an axis: accumulates the lag products and the 2-D histogram of (s[i], s[i+1]) */
static void feed_sequence(const uint32_t *s, int len, Moments *m, uint32_t *hist_top, uint32_t *hist_bot, uint32_t *hist1)
{
    for (int i = 0; i < len; ++i) {
        const double u = ((double)s[i] + 0.5) * (1.0 / 4294967296.0) - 0.5;
        m->sx += u; m->sxx += u * u; m->n += 1.0;
        m->pop += (double)__builtin_popcount(s[i]);
        hist1[s[i] >> 24] += 1u;
        for (int lag = 1; lag <= 4 && i + lag < len; ++lag) {
            const double w = ((double)s[i + lag] + 0.5) * (1.0 / 4294967296.0) - 0.5;
            m->sxy[lag - 1] += u * w; m->npair[lag - 1] += 1.0;
        }
        if (i + 1 < len) {
            hist_top[((s[i] >> 24) << 8) | (s[i + 1] >> 24)] += 1u;
            hist_bot[((s[i] & 255u) << 8) | (s[i + 1] & 255u)] += 1u;
        }
    }
}

static double chi2_z(const uint32_t *hist, int cells, double total)
{
    const double e = total / (double)cells;
    double c = 0.0;
    for (int i = 0; i < cells; ++i) { const double d = (double)hist[i] - e; c += d * d / e; }
    const double dof = (double)(cells - 1);
    return (c - dof) / sqrt(2.0 * dof);
}

/* Runs the battery with `rounds` Philox rounds; `calls_per_axis` Philox calls are drawn per axis (sequence length 64 along
 * the axis, as many sequences as that gives).  stats[AX_COUNT][STAT_PER_AXIS] z-scores; pair_stats[4][2] = chi-square z
 * (top bytes, bottom bytes) between the word positions (0,1), (1,2), (2,3), (0,3) of one call; derived[3] = z of the
 * click-vs-competitor-uniform independence chi-square (16 x 2 table ... reported as z), z of the click frequency, and the
 * z of the competitor uniform's top-byte uniformity.  Returns the number of words drawn. */
BAT_API double bat_run(int rounds, int64_t calls_per_axis, uint64_t seed, double *stats, double *pair_stats, double *derived)
{
    enum { LEN = 64 };
    const int64_t nseq = calls_per_axis / LEN;
    double words = 0.0;
    for (int ax = 0; ax < AX_COUNT; ++ax) {
        Moments tot;
        memset(&tot, 0, sizeof tot);
        uint32_t *H_top = calloc(65536, 4), *H_bot = calloc(65536, 4), *H1 = calloc(256, 4);
        /* (positions 0..3 of one call against each other: only on the AX_WORDS pass) */
        uint32_t *P_top = calloc(4 * 65536, 4), *P_bot = calloc(4 * 65536, 4);
#pragma omp parallel
        {
            Moments m;
            memset(&m, 0, sizeof m);
            uint32_t *h_top = calloc(65536, 4), *h_bot = calloc(65536, 4), *h1 = calloc(256, 4);
            uint32_t *p_top = calloc(4 * 65536, 4), *p_bot = calloc(4 * 65536, 4);
#pragma omp for schedule(static)
            for (int64_t q = 0; q < nseq; ++q) {
                /* where this sequence starts: a pseudo-random but realistic point of the counter space */
                const uint64_t h = splitmix64(seed ^ (uint64_t)(q * 6 + ax));
                const uint32_t env0 = (uint32_t)(h & 0xFFFFF), kw0 = (uint32_t)((h >> 20) & 0xFFF), tick0 = (uint32_t)((h >> 32) & 0xFFFF);
                const uint32_t idx0 = (uint32_t)((h >> 48) & 0x3FF);
                uint32_t s[4][LEN];
                for (int i = 0; i < LEN; ++i) {
                    uint32_t idx = idx0, stage = 1u /* ST_AUCTION */, kw = kw0, tick = tick0;
                    uint64_t env = env0;
                    if (ax == AX_INDEX || ax == AX_WORDS) idx += (uint32_t)i;
                    else if (ax == AX_KEYWORD) kw += (uint32_t)i;
                    else if (ax == AX_TICK) tick += (uint32_t)i;
                    else if (ax == AX_ENV) env += (uint64_t)i;
                    else if (ax == AX_VOLUME) { stage = 0u /* ST_VOL */; idx = 0u; kw += (uint32_t)i; }      /* call (0, VOL, k/4): consecutive k/4 */
                    const uint64_t key = env_key(seed, env);
                    uint32_t w[4];
                    philox(idx, stage, kw, tick, (uint32_t)key, (uint32_t)(key >> 32), rounds, w);
                    for (int c = 0; c < 4; ++c) s[c][i] = w[c];
                }
                if (ax == AX_WORDS) {
                    /* the stream of auction words in auction order: j = 4 index + position */
                    uint32_t flat[4 * LEN];
                    for (int i = 0; i < LEN; ++i)
                        for (int c = 0; c < 4; ++c) flat[4 * i + c] = s[c][i];
                    feed_sequence(flat, 4 * LEN, &m, h_top, h_bot, h1);
                    static const int pa[4] = {0, 1, 2, 0}, pb[4] = {1, 2, 3, 3};
                    for (int i = 0; i < LEN; ++i)
                        for (int p = 0; p < 4; ++p) {
                            const uint32_t a = s[pa[p]][i], b = s[pb[p]][i];
                            p_top[p * 65536 + (((a >> 24) << 8) | (b >> 24))] += 1u;
                            p_bot[p * 65536 + (((a & 255u) << 8) | (b & 255u))] += 1u;
                        }
                } else {
                    for (int c = 0; c < 4; ++c) feed_sequence(s[c], LEN, &m, h_top, h_bot, h1);
                }
            }
#pragma omp critical
            {
                tot.sx += m.sx; tot.sxx += m.sxx; tot.n += m.n; tot.pop += m.pop;
                for (int l = 0; l < 4; ++l) { tot.sxy[l] += m.sxy[l]; tot.npair[l] += m.npair[l]; }
                for (int i = 0; i < 65536; ++i) { H_top[i] += h_top[i]; H_bot[i] += h_bot[i]; }
                for (int i = 0; i < 256; ++i) H1[i] += h1[i];
                for (int i = 0; i < 4 * 65536; ++i) { P_top[i] += p_top[i]; P_bot[i] += p_bot[i]; }
            }
            free(h_top); free(h_bot); free(h1); free(p_top); free(p_bot);
        }
        double *st = stats + ax * STAT_PER_AXIS;
        const double var = 1.0 / 12.0;
        for (int l = 0; l < 4; ++l) st[l] = (tot.sxy[l] / tot.npair[l]) / var * sqrt(tot.npair[l]);      /* r sqrt(n) ~ N(0,1) */
        double pairs = 0.0;
        for (int i = 0; i < 65536; ++i) pairs += (double)H_top[i];
        st[4] = chi2_z(H_top, 65536, pairs);
        st[5] = chi2_z(H_bot, 65536, pairs);
        st[6] = chi2_z(H1, 256, tot.n);
        st[7] = (tot.pop / tot.n - 16.0) / sqrt(8.0 / tot.n);          /* popcount of a uniform word: mean 16, variance 8 */
        if (ax == AX_WORDS)
            for (int p = 0; p < 4; ++p) {
                double t = 0.0;
                for (int i = 0; i < 65536; ++i) t += (double)P_top[p * 65536 + i];
                pair_stats[2 * p] = chi2_z(P_top + p * 65536, 65536, t);
                pair_stats[2 * p + 1] = chi2_z(P_bot + p * 65536, 65536, t);
            }
        words += tot.n;
        free(H_top); free(H_bot); free(H1); free(P_top); free(P_bot);
    }
    /* the derived pair of one auction word (oracle/adcraft_oracle.c orc_auction_outcome, adc_law.h auction_uniform24): click
     * = word < T; inside either outcome the word's offset rescaled to 24 bits is the competitor's uniform.  Independence of
     * the two on PHILOX words, for a click rate of 0.37: 2 x 4096 contingency table on the uniform's top 12 bits. */
    {
        const double ctr = 0.37;
        const uint32_t T = (uint32_t)(ctr * 4294967296.0);
        const int64_t ncalls = calls_per_axis;
        uint32_t *tab = calloc(2 * 4096, 4);
        double nclick = 0.0, n = 0.0;
#pragma omp parallel
        {
            uint32_t *t = calloc(2 * 4096, 4);
            double nc = 0.0, nn = 0.0;
#pragma omp for schedule(static)
            for (int64_t q = 0; q < ncalls; ++q) {
                const uint64_t h = splitmix64(seed ^ 0xD1CEull ^ (uint64_t)q);
                const uint64_t key = env_key(seed, h & 0xFFFFF);
                uint32_t w[4];
                philox((uint32_t)(q & 0xFF), 1u, (uint32_t)((h >> 20) & 0xFFF), (uint32_t)((h >> 32) & 0xFFFF), (uint32_t)key, (uint32_t)(key >> 32), rounds, w);
                for (int c = 0; c < 4; ++c) {
                    const int click = w[c] < T;
                    const uint32_t d = click ? w[c] : w[c] - T;
                    const uint64_t range = click ? (uint64_t)T : 4294967296ull - T;
                    uint32_t v = (uint32_t)(((unsigned __int128)d << 24) / range);          /* the exact rescaling; the law's differs by < 1 word per value */
                    if (v > 0xFFFFFFu) v = 0xFFFFFFu;
                    t[(click << 12) | (v >> 12)] += 1u;
                    nc += click; nn += 1.0;
                }
            }
#pragma omp critical
            {
                for (int i = 0; i < 2 * 4096; ++i) tab[i] += t[i];
                nclick += nc; n += nn;
            }
            free(t);
        }
        /* independence chi-square of the 2 x 4096 table (dof 4095) */
        double c2 = 0.0;
        const double pc[2] = {1.0 - nclick / n, nclick / n};
        for (int b = 0; b < 4096; ++b) {
            const double col = (double)tab[b] + (double)tab[4096 + b];
            for (int k = 0; k < 2; ++k) {
                const double e = col * pc[k], d = (double)tab[(k << 12) | b] - e;
                c2 += d * d / e;
            }
        }
        derived[0] = (c2 - 4095.0) / sqrt(2.0 * 4095.0);
        derived[1] = (nclick - n * ((double)T / 4294967296.0)) / sqrt(n * ctr * (1.0 - ctr));
        uint32_t *marg = calloc(4096, 4);
        for (int b = 0; b < 4096; ++b) marg[b] = tab[b] + tab[4096 + b];
        derived[2] = chi2_z(marg, 4096, n);
        free(marg); free(tab);
        words += n;
    }
    return words;
}

/* the same Philox as a table, for the GPU mirror of the known-answer check (tests compare a device-side draw with this) */
BAT_API void bat_philox(const uint32_t *ctr4, const uint32_t *key2, int64_t n, int rounds, uint32_t *out4)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) philox(ctr4[4 * i], ctr4[4 * i + 1], ctr4[4 * i + 2], ctr4[4 * i + 3], key2[2 * i], key2[2 * i + 1], rounds, out4 + 4 * i);
}
