/*
 * adcraft_oracle.c - CPU ORACLE for the BiddingSimulation step path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (adcraft_amd/) never links,
 * imports or calls anything in this directory and has no CPU fallback.
 *
 * It restates, in plain C, the reference's per-step algorithm literally - the same loops
 * in the same order - and cites the reference lines each function follows:
 *
 *   orc_nth_price_auction        adcraft/synthetic_kw_helpers.py:116-180
 *   cell walk (campaign loop)    adcraft/bidding_simulation.py:170-234
 *   one (t,k) cell               adcraft/bidding_simulation.py:44-120
 *   24-way volume split          adcraft/bidding_simulation.py:151-167
 *   step tail                    adcraft/gymnasium_kw_env.py:197-244
 *   drift                        adcraft/gymnasium_kw_env.py:114-158
 *   EXPLICIT cell (+phantom)     adcraft/synthetic_kw_classes.py:493-538, src/lib.rs:54-76,93-105
 *   volume                       src/lib.rs:314-325
 *
 * Two variate sources:
 *   TAPE   - the variates are supplied by the caller in the order the reference draws them
 *            (competitor bids, click booleans, conversion booleans, revenues ...).  Pinned
 *            bit-exactly against tests/golden/g3_*.json and g8_*.json, which were recorded
 *            from the reference itself (tools/gen_golden.py).
 *   PHILOX - the production stream (revision 2): Philox4x32-7 addressed by (env key; auction index, stage,
 *            keyword, tick), with float32 transforms built only from IEEE-exact operations
 *            (+ - * / sqrt fma rint), so a GPU computes bit-identical values.  The Philox core is
 *            pinned by the Random123 known-answer vectors, the transforms by closed forms
 *            (tests/test_oracle_*.py).
 *
 * Parity status: pinned (TAPE mode against reference-generated fixtures).  The reference's
 * own step-time random stream (numpy PCG64 interleaved with unseeded Rust thread_rng,
 * src/lib.rs:25,43,61,75,320) is not reproducible even by the reference; PHILOX mode is
 * this project's stream and is checked distributionally against the reference's laws.
 *
 * Money: IMPLICIT bids, competitor bids and revenues are whole cents in the reference
 * (gymnasium_kw_env.py:215, synthetic_kw_helpers.py:68-70,108-113), so the oracle carries
 * integer cents.  EXPLICIT costs are un-rounded reals (src/lib.rs:60-64) and are carried as
 * doubles in the reference's own operation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

enum { ORC_IMPLICIT = 0, ORC_EXPLICIT = 1, ORC_IMPLICIT_GENERAL = 2 };
enum { P_VOL_MEAN = 0, P_VOL_STD, P_A, P_B, P_BCTR, P_SCTR, P_REV_MEAN, P_REV_STD, P_COUNT };
/* P_A / P_B: IMPLICIT cost_loc / cost_scale (Laplace), EXPLICIT imp_intercept / imp_slope */
enum { ST_VOL = 0, ST_AUCTION = 1, ST_DRIFT = 2, ST_XPHANTOM = 3, ST_XREV = 4, ST_ACTION = 5, ST_METRIC = 6, ST_CONV = 7, ST_KEYGEN = 8,
       ST_AGENT = 9, ST_GBIDDERS = 10, ST_GBID = 11, ST_GCLICK = 12 };
/* IMPLICIT_GENERAL (the reference's default ImplicitKeyword; stream revisions 3 and 4 for this model; revision 4: call (t/4,
 * ST_GBIDDERS) word t%4 = the uniform the bidder count of sub-timestep t is read off, orc_bidders_from_word; where that is not
 * applicable, as in revision 3:) call (64 t + b/4, ST_GBIDDERS)
 * word b%4 = participation coin of bidder b in sub-timestep t; call (j, ST_GBID) = the exponential spacings of the top bids of
 * auction j, highest first (orc_top_laplace_bids); call (j, ST_GCLICK) = {click, conversion, revenue} words of auction j. */
/* Stream layout (revision 2): call (0, ST_VOL, k/4) holds the volume words of keywords 4(k/4)..+3 (word k%4).
 * IMPLICIT: call (j/4, ST_AUCTION) holds one word per auction j (word j%4); that word decides the click (word < T) and,
 * rescaled inside its sub-interval, is the competitor-bid uniform (orc_auction_outcome); the word 2^32-1 never wins;
 * call (j, ST_CONV) holds {conversion, revenue} words (x,y) of auction j, only consumed for a paid click. */
#define ORC_TIMESTEPS 24
#define ORC_VMAX (1 << 20)

/* ------------------------------------------------------------------ Philox4x32-R */
/* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC11); constants of Random123.  The stream uses R = 7
 * (the paper's Crush-resistant round count; 10 is its default with a safety margin); both are pinned by the Random123
 * known-answer vectors in tests/test_oracle_scalar.py. */
#define ORC_PHILOX_ROUNDS 7
ORC_API void orc_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
ORC_API void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    orc_philox4x32_r(ctr, key, ORC_PHILOX_ROUNDS, out);
}

static void draw(uint64_t key, uint32_t index, uint32_t stage, uint32_t kw, uint32_t tick, uint32_t w[4])
{
    uint32_t c[4] = { index, stage, kw, tick };
    uint32_t k[2] = { (uint32_t)key, (uint32_t)(key >> 32) };
    orc_philox4x32(c, k, w);
}

/* ------------------------------------------------------------------ deterministic f32 math */
/* Every step is one correctly-rounded IEEE operation, written out explicitly (no contraction:
 * build with -ffp-contract=off), so any conforming CPU or GPU produces the same bits. */
static float as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* natural log of a positive normal float; Cephes logf scheme */
ORC_API float orc_det_logf(float x)
{
    static const float P[9] = { 7.0376836292E-2f, -1.1514610310E-1f, 1.1676998740E-1f, -1.2420140846E-1f,
                                1.4249322787E-1f, -1.6668057665E-1f, 2.0000714765E-1f, -2.4999993993E-1f,
                                3.3333331174E-1f };
    uint32_t u = as_u32(x);
    int e = (int)(u >> 23) - 126;                          /* x = m * 2^e, m in [0.5,1) */
    float m = as_f32((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }
    float f = m - 1.0f;
    float z = f * f;
    float p = P[0];
    for (int i = 1; i < 9; ++i) p = fmaf(p, f, P[i]);
    float y = f * (z * p);
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* e^x for |x| <= 87; Cephes expf scheme */
ORC_API float orc_det_expf(float x)
{
    if (x > 87.0f) x = 87.0f;
    if (x < -87.0f) x = -87.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float y = fmaf(p, z, r) + 1.0f;
    int ni = (int)n;
    return y * as_f32((uint32_t)(ni + 127) << 23);
}

/* uniform strictly inside (0,1) from the top 23 bits: (i + 1/2) / 2^23, exact in f32 */
static float u23(uint32_t w) { return ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-07f; }
/* uniform in [0,1) from the top 24 bits */
static float u24(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }

/* standard normal from one word: sign bit 8, lower-tail probability from bits 31..9;
 * Wichura's AS241 PPND7 rational approximations (Appl. Statist. 37 (1988) 477-484). */
ORC_API float orc_normal_from_word(uint32_t w)
{
    float p = ((float)(w >> 9) + 0.5f) * 5.9604644775390625e-08f;   /* (0, 0.5) */
    float q = p - 0.5f;
    float val;
    if (q >= -0.425f) {
        float r = fmaf(-q, q, 0.180625f);
        float num = fmaf(fmaf(fmaf(5.9109374720e1f, r, 1.5929113202e2f), r, 5.0434271938e1f), r, 3.3871327179e0f);
        float den = fmaf(fmaf(fmaf(6.7187563600e1f, r, 7.8757757664e1f), r, 1.7895169469e1f), r, 1.0f);
        val = q * num / den;                                          /* <= 0 */
    } else {
        float r = sqrtf(-orc_det_logf(p)) - 1.6f;
        float num = fmaf(fmaf(fmaf(1.7023821103e-1f, r, 1.3067284816e0f), r, 2.7568153900e0f), r, 1.4234372777e0f);
        float den = fmaf(fmaf(1.2021132975e-1f, r, 7.3700164250e-1f), r, 1.0f);
        val = -(num / den);
    }
    return ((w >> 8) & 1u) ? -val : val;
}

/* competitor bid in cents: round2(max(|Laplace(loc, scale)|, 0))  (synthetic_kw_helpers.py:104-113)
 * Laplace(0,1) = random sign x Exponential(1). */
ORC_API int32_t orc_laplace_cents_from_word(uint32_t w, float loc, float scale)
{
    float e = -orc_det_logf(u23(w));
    float z = ((w >> 8) & 1u) ? e : -e;
    float a = fabsf(fmaf(scale, z, loc));
    float c = rintf(a * 100.0f);
    if (!(c < 1.0e9f)) c = 1.0e9f;
    return (int32_t)c;
}

/* Bernoulli threshold: event <=> (uint64)word < T, T = round(p * 2^32) in [0, 2^32]
 * (reference: rng.random(n) <= p, synthetic_kw_helpers.py:73-77) */
ORC_API uint64_t orc_bernoulli_threshold(float p)
{
    double t = floor((double)p * 4294967296.0 + 0.5);
    if (!(t > 0.0)) return 0;
    if (t > 4294967296.0) t = 4294967296.0;
    return (uint64_t)t;
}

/* one word per auction: click <=> word < T (rng.random() <= buyside_ctr, synthetic_kw_helpers.py:73-77); the word is
 * uniform inside either sub-interval, so d / range is a uniform independent of the click; 24 bits of it feed the
 * competitor bid round2(|Laplace(loc, scale)|) (synthetic_kw_helpers.py:104-113): bit 0 sign, bits 23..1 magnitude. */
/* -log(u), u = w24 / 2^24 (w24 odd): exponent from the float bits, mantissa through a 256-interval table of
 * orc_det_logf on [1,2) with linear interpolation; the table is built from orc_det_logf itself */
static float g_log_value[257], g_log_slope[256];
static int g_log_ready = 0;
static void build_log_table(void)
{
    for (int i = 0; i <= 256; ++i) g_log_value[i] = orc_det_logf(1.0f + (float)i * 0.00390625f);
    for (int i = 0; i < 256; ++i) g_log_slope[i] = (g_log_value[i + 1] - g_log_value[i]) * 3.0517578125e-05f;
    g_log_ready = 1;
}
ORC_API float orc_neg_log_u24(uint32_t w24)
{
    if (!g_log_ready) build_log_table();
    const uint32_t bits = as_u32((float)w24);
    const float ef = (float)((int)(bits >> 23) - 151);
    const uint32_t mant = bits & 0x007FFFFFu;
    const uint32_t i = mant >> 15;
    float r = fmaf(g_log_slope[i], (float)(mant & 0x7FFFu), g_log_value[i]);
    r = fmaf(ef, -2.12194440e-4f, r);
    r = fmaf(ef, 0.693359375f, r);
    return -r;
}

/* Standard normal from ONE word by a table of the inverse CDF (revision 2).  bit 31 = sign, bits 30..8 = m, t = 2m + 1,
 * lower-tail probability p = t / 2^25 in (0, 1/2); |z| = -Phi^-1(p) is interpolated linearly inside one of 24 x 32
 * intervals addressed by the float representation of t (exponent, top 5 mantissa bits).  Nodes: AS241 PPND7 (the code of
 * orc_normal_from_word), so the table holds the same bits wherever it is built. */
#define ORC_NORM_ENTRIES 768
static float g_norm_value[ORC_NORM_ENTRIES + 1], g_norm_slope[ORC_NORM_ENTRIES];
static int g_norm_ready = 0;
static float tail_magnitude(float p)            /* -Phi^-1(p), 0 < p <= 1/2 */
{
    float q = p - 0.5f;
    if (q >= -0.425f) {
        float r = fmaf(-q, q, 0.180625f);
        float num = fmaf(fmaf(fmaf(5.9109374720e1f, r, 1.5929113202e2f), r, 5.0434271938e1f), r, 3.3871327179e0f);
        float den = fmaf(fmaf(fmaf(6.7187563600e1f, r, 7.8757757664e1f), r, 1.7895169469e1f), r, 1.0f);
        return -(q * num / den);
    } else {
        float r = sqrtf(-orc_det_logf(p)) - 1.6f;
        float num = fmaf(fmaf(fmaf(1.7023821103e-1f, r, 1.3067284816e0f), r, 2.7568153900e0f), r, 1.4234372777e0f);
        float den = fmaf(fmaf(1.2021132975e-1f, r, 7.3700164250e-1f), r, 1.0f);
        return num / den;
    }
}
static void build_norm_table(void)
{
    for (int i = 0; i < ORC_NORM_ENTRIES; ++i) {
        float t0 = ldexpf(1.0f + (float)(i & 31) * 0.03125f, i >> 5);      /* left end of interval i, in units of t */
        g_norm_value[i] = tail_magnitude(t0 * 2.98023223876953125e-08f);   /* p = t0 / 2^25 */
    }
    g_norm_value[ORC_NORM_ENTRIES] = 0.0f;                                 /* p = 1/2 */
    for (int i = 0; i < ORC_NORM_ENTRIES; ++i) g_norm_slope[i] = (g_norm_value[i + 1] - g_norm_value[i]) * 3.814697265625e-06f;
    g_norm_ready = 1;
}
ORC_API float orc_normal_tab(uint32_t w)
{
    if (!g_norm_ready) build_norm_table();
    const uint32_t t = 2u * ((w >> 8) & 0x007FFFFFu) + 1u;
    const uint32_t bits = as_u32((float)t);
    const uint32_t i = (bits >> 18) - (127u << 5);
    const float z = fmaf(g_norm_slope[i], (float)(bits & 0x0003FFFFu), g_norm_value[i]);
    return (w >> 31) ? z : -z;
}
/* IMPLICIT revenue in cents: round2(max(N(mu, sd), 0.01))  (synthetic_kw_helpers.py:66-70) */
ORC_API int32_t orc_revenue_cents_tab(uint32_t w, float mu, float sd)
{
    float x = fmaf(sd, orc_normal_tab(w), mu);
    x = fmaxf(x, 0.01f);
    float c = rintf(x * 100.0f);
    if (!(c < 1.0e9f)) c = 1.0e9f;
    return (int32_t)c;
}

static uint32_t rescale_multiplier(uint64_t range)      /* floor(2^56 / range) in float32, saturated */
{
    if (range == 0) return 0u;
    float m = floorf(72057594037927936.0f / (float)range);
    if (!(m < 4294967040.0f)) m = 4294967040.0f;
    return (uint32_t)m;
}
/* competitor bid in cents from the auction's 24-bit uniform v.  Layout (revision 2): v < 2^23 is the negative side of the
 * Laplace deviate, z = -e(v); v >= 2^23 the positive side, z = +e(2^24 - 1 - v); e(m) = -log((2m + 1) / 2^24) by the
 * table.  loc + |scale| z is non-decreasing in v. */
ORC_API int32_t orc_competitor_cents_from_v(uint32_t v, float loc, float scale)
{
    float z;
    if (v < 0x00800000u) z = -orc_neg_log_u24(2u * v + 1u);
    else z = orc_neg_log_u24(2u * (0x00FFFFFFu - v) + 1u);
    const float a = fabsf(fmaf(fabsf(scale), z, loc));
    float c = rintf(a * 100.0f);
    if (!(c < 1.0e9f)) c = 1.0e9f;
    return (int32_t)c;
}
/* checker: number of v in [1, 2^24) at which the Laplace deviate z(v) DEcreases (must be 0: the thresholds of
 * k_step_implicit_fast rest on z, hence loc + |scale| z and its rounding to cents, being monotone in v) */
ORC_API int64_t orc_check_deviate_monotone(void)
{
    int64_t bad = 0;
    float prev = -orc_neg_log_u24(1u);
    for (uint32_t v = 1; v < 0x01000000u; ++v) {
        const float z = v < 0x00800000u ? -orc_neg_log_u24(2u * v + 1u) : orc_neg_log_u24(2u * (0x00FFFFFFu - v) + 1u);
        if (!(z >= prev)) ++bad;
        prev = z;
    }
    return bad;
}
ORC_API int32_t orc_auction_outcome(uint32_t w, float bctr, float loc, float scale, int32_t *click_out)
{
    const uint64_t T = orc_bernoulli_threshold(bctr);
    const int click = (uint64_t)w < T;
    const uint32_t d = click ? w : w - (uint32_t)T;
    const uint32_t m = click ? rescale_multiplier(T) : rescale_multiplier(4294967296ull - T);
    uint32_t v = (uint32_t)(((uint64_t)d * m) >> 32);     /* ~ floor(d * 2^24 / range): integer multiply */
    if (v > 0x00FFFFFFu) v = 0x00FFFFFFu;
    *click_out = click;
    return orc_competitor_cents_from_v(v, loc, scale);
}

/* revenue in cents: round2(max(N(mu, sd), 0.01))  (synthetic_kw_helpers.py:66-70) */
ORC_API int32_t orc_revenue_cents_from_word(uint32_t w, float mu, float sd)
{
    float x = fmaf(sd, orc_normal_from_word(w), mu);
    x = fmaxf(x, 0.01f);
    float c = rintf(x * 100.0f);
    if (!(c < 1.0e9f)) c = 1.0e9f;
    return (int32_t)c;
}

/* volume: round_half_away(max(N(mean, sd), 0))  (src/lib.rs:314-325) */
ORC_API int32_t orc_volume_from_word(uint32_t w, float mean, float sd)
{
    float x = fmaf(sd, orc_normal_tab(w), mean);
    x = fmaxf(x, 0.0f);
    if (!(x < (float)ORC_VMAX)) x = (float)ORC_VMAX;
    float t = truncf(x);
    int32_t v = (int32_t)t + ((x - t) >= 0.5f ? 1 : 0);
    return v;
}

/* bid / budget canonicalisation (gymnasium_kw_env.py:199,215): numpy round(x,2) = rint(x*100)/100 in f64 */
ORC_API int64_t orc_bid_cents(float bid)
{
    double c = rint((double)bid * 100.0);
    if (!(c >= 1.0)) c = 1.0;
    if (c > 1.0e9) c = 1.0e9;
    return (int64_t)c;
}
/* checker for the product's division-free cents -> dollars (adcraft_amd/csrc/adc_law.h:cents_to_dollars_f64): counts
 * the cents in [lo, hi) for which q0 = c*0.01; q = fma(fma(-q0,100,c), 0.01, q0) differs from the IEEE quotient c/100.0
 * the reference computes (bidding_simulation.py:97-104 works on np.round(cost, 2) dollars). */
ORC_API int64_t orc_check_div100(int64_t lo, int64_t hi)
{
    int64_t bad = 0;
    for (int64_t c = lo; c < hi; ++c) {
        const double a = (double)c, q0 = a * 0.01;
        const double q = fma(fma(-q0, 100.0, a), 0.01, q0);
        if (q != a / 100.0) ++bad;
    }
    return bad;
}

/* the same for float32: q0 = c*0.01f; q = fmaf(fmaf(-q0,100,c), 0.01f, q0) vs (float)c / 100.0f, for |c| in [lo, hi) */
ORC_API int64_t orc_check_div100f(int64_t lo, int64_t hi)
{
    int64_t bad = 0;
    for (int64_t c = lo; c < hi; ++c) {
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            const float a = (float)(sgn * c), q0 = a * 0.01f;
            const float q = fmaf(fmaf(-q0, 100.0f, a), 0.01f, q0);
            if (q != a / 100.0f) ++bad;
        }
    }
    return bad;
}

ORC_API int64_t orc_budget_cents(float budget)
{
    double c = rint((double)budget * 100.0);
    if (!(c > -9.0e15)) c = -9.0e15;
    if (c > 9.0e15) c = 9.0e15;
    return (int64_t)c;
}

/* EXPLICIT impression probability (src/lib.rs:93-105, 290-300), float32 form used on the step path */
ORC_API float orc_threshold_sigmoid_f32(float bid, float thresh, float intercept, float slope)
{
    float th = 2.0f * thresh;            /* halver = 2 + 1e-10 == 2.0f in float32 */
    th = fminf(fmaxf(th, 0.0f), 1.0f) / 2.0f;
    float r = 1.0f / (1.0f + orc_det_expf(-slope * (bid - intercept)));
    float v = fmaf(fmaf(2.0f, th, 1.0f), r, -th);
    return fminf(fmaxf(v, 0.0f), 1.0f);
}
/* the same in f64 exactly as the Rust source computes it (scalar FFI shim parity) */
ORC_API double orc_threshold_sigmoid_f64(double p, double thresh, double intercept, double slope)
{
    double halver = 2.0 + 1e-10;
    double t = halver * thresh;
    t = (t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t)) / halver;
    double r = 1.0 / (1.0 + exp(-slope * (p - intercept)));
    double v = (1.0 + 2.0 * t) * r - t;
    return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
}
ORC_API double orc_sigmoid_f64(double x, double s, double t) { return 1.0 / (1.0 + exp(-s * (x - t))); }

/* EXPLICIT per-impression cost (src/lib.rs:54-67): clamp(sqrt(x)/4 + 4.4/2 + N(0, 1e-10 + sqrt(x)/6), 0, 4.4) */
ORC_API float orc_explicit_cost_from_word(uint32_t w, float bid)
{
    float sq = sqrtf(bid);
    float sd = 1e-10f + sq / 6.0f;
    float v = fmaf(sd, orc_normal_from_word(w), sq / 4.0f + 2.2f);
    return fminf(fmaxf(v, 0.0f), 4.4f);
}

/* IMPLICIT_GENERAL, stream revision 4: the number of bidders of a call, Binomial(max_bidders, rate) (synthetic_kw_classes.py:610-621:
 * rng.random(n) <= rate, of which only the count is used), read off one uniform by walking the pmf up from 0 in float32:
 * pmf(0) = q^n, pmf(b + 1) = pmf(b) (n - b) / (b + 1) p / q.  pmf0 = 0: not applicable (rate outside (0, 1) or q^n too small) -
 * then the max_bidders coins of revision 3 are drawn. */
ORC_API void orc_bidder_law(int32_t max_bidders, float rate, float *pmf0, float *ratio)
{
    *pmf0 = 0.0f;
    *ratio = 0.0f;
    if (!(rate > 0.0f && rate < 1.0f) || max_bidders < 1 || max_bidders >= 128) return;      /* (the engine keeps the sums in a table of 128) */
    const float q = 1.0f - rate;
    const float lp = (float)max_bidders * orc_det_logf(q);
    if (!(lp > -60.0f)) return;
    *pmf0 = orc_det_expf(lp);
    *ratio = rate / q;
}
ORC_API int32_t orc_bidders_from_word(uint32_t w, int32_t n, float pmf0, float ratio)
{
    const float u = u23(w);
    float pmf = pmf0, cdf = pmf0;
    int32_t b = 0;
    while (u > cdf && b < n) {
        pmf = pmf * ((float)(n - b) / (float)(b + 1)) * ratio;
        cdf = cdf + pmf;
        ++b;
    }
    return b;
}

/* IMPLICIT_GENERAL, stream revision 3: the top k = min(B, top) of B iid Laplace(loc, |scale|) bids (rng.laplace,
 * adcraft/synthetic_kw_classes.py:681-686), descending, as order statistics: t_1 = E_1 / B, t_2 = t_1 + E_2 / (B - 1), ... with
 * E_i = -log(u_i) are -log of the top uniform order statistics (Renyi); the Laplace quantile at p = exp(-t) is
 * loc + s (ln 2 - t) for p <= 1/2 and loc - s log(2 (1 - p)) above.  nth_price_auction only looks at the top (w + n). */
ORC_API int32_t orc_top_laplace_bids(const uint32_t w[4], int32_t B, int32_t top, float loc, float scale, float out[4])
{
    const int32_t k = B < top ? B : top;
    const float s = fabsf(scale);
    float t = 0.0f;
    for (int32_t i = 0; i < k && i < 4; ++i) {
        const float e = -orc_det_logf(u23(w[i]));
        t = t + e / (float)(B - i);
        float z;
        if (t >= 0.693147182464599609375f) z = 0.693147182464599609375f - t;
        else {
            float q = 1.0f - orc_det_expf(-t);
            q = q > 5.9604644775390625e-08f ? q : 5.9604644775390625e-08f;
            z = -orc_det_logf(q + q);
        }
        out[i] = fmaf(s, z, loc);
    }
    return k;
}

/* ------------------------------------------------------------------ nth_price_auction */
/* synthetic_kw_helpers.py:116-180.  other_bids is [n_auctions][n_bidders] row-major.
 * Returns impressions; fills placements/costs (length = impressions). */
static int cmp_f64(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}
ORC_API int32_t orc_nth_price_auction(double bid, const double *other_bids, int32_t n_auctions, int32_t n_bidders,
                                      int32_t n, int32_t num_winners, int32_t *placements, double *costs)
{
    int32_t top = num_winners + n, impressions = 0;
    int32_t width = n_bidders > top ? n_bidders : top;
    double *row = (double *)malloc(sizeof(double) * (size_t)(width > 0 ? width : 1));
    for (int32_t a = 0; a < n_auctions; ++a) {
        const double *src = other_bids + (size_t)a * n_bidders;
        const double *top_n;
        if (n_bidders >= top) {                  /* :152-155 top (w+n) bids, ascending */
            memcpy(row, src, sizeof(double) * n_bidders);
            qsort(row, n_bidders, sizeof(double), cmp_f64);
            top_n = row + (n_bidders - top);
        } else {                                 /* :156-161 pad with zero bids, then sort */
            for (int32_t i = 0; i < top - n_bidders; ++i) row[i] = 0.0;
            memcpy(row + (top - n_bidders), src, sizeof(double) * n_bidders);
            qsort(row, top, sizeof(double), cmp_f64);
            top_n = row;
        }
        int32_t index = 0;                       /* :167 searchsorted(auction, bid), side="left" */
        while (index < top && top_n[index] < bid) ++index;
        if (index > n) {                         /* :170 */
            placements[impressions] = top - index;
            if (n > 1) {
                int32_t ci = index - (n - 1);
                if (ci < 0) ci = 0;
                costs[impressions] = top_n[ci];
            } else {
                costs[impressions] = bid;
            }
            ++impressions;
        }
    }
    free(row);
    return impressions;
}

/* ------------------------------------------------------------------ the step */
typedef struct {
    int32_t num_envs, num_keywords, model, max_days;
    double loss_threshold;
    float drift_vol, drift_ctr, drift_cvr;
    int32_t drift_on;        /* updater_mask all-True */
    float imp_thresh;        /* EXPLICIT impression_thresh (0.05 in the env, gymnasium_kw_utils.py:81) */
    int32_t auto_reset;      /* vector form: done envs restart (day=0, cum=0) after reporting */
    int32_t threads;         /* OpenMP threads over envs (cpu_baseline); 0/1 = serial */
    /* IMPLICIT_GENERAL: the bidder pool of ImplicitKeyword._bidder_distribution_init (synthetic_kw_classes.py:659-665) and the
     * number of winning placements of the auction (ImplicitKeyword.auction's n_winners, :623) */
    int32_t max_bidders;     /* default 30 */
    float participation_rate;/* default 0.6 */
    int32_t num_winners;     /* default 1 */
} orc_config;

typedef struct {             /* TAPE source: flat per-call tapes + cursors (advanced by the step) */
    const int32_t *volumes;      /* [N][K] */
    const int32_t *bid_cents;    /* IMPLICIT: competitor bids, n per visited cell */
    const int32_t *x_impressions;/* EXPLICIT: Binomial result per visited cell; IMPLICIT_GENERAL: bidders of the cell */
    const double *x_cost;        /* EXPLICIT: per-impression costs; IMPLICIT_GENERAL: bids, bidders x auctions per cell */
    const uint8_t *click, *conv;
    const int32_t *rev_cents;
    int64_t cur_bid, cur_ximp, cur_xcost, cur_click, cur_conv, cur_rev;  /* in/out */
    const float *drift_uniforms; /* nullable [3][N][K]: the vectors np_random.uniform(-a, a, size=K) of update_keywords()
                                  * in its order vol, ctr, cvr (gymnasium_kw_env.py:132-135); applied at the end of the step */
} orc_tape;

typedef struct {
    int32_t *impressions, *clicks, *conversions;   /* [N][K] */
    int64_t *cost_cents, *revenue_cents;           /* [N][K] IMPLICIT exact money; EXPLICIT revenue only */
    double *cost, *revenue;                        /* [N][K] dollars (IMPLICIT: cents/100) */
    int32_t *volumes;                              /* [N][K] V drawn this step */
    double *reward, *cum_profit;                   /* [N] */
    int32_t *day;                                  /* [N] days_passed after the step */
    uint8_t *terminated, *truncated;               /* [N] */
} orc_out;

/* the combined BiddingOutcomes of the step (bidding_simulation.py:10-38), optional: every paid click in the reference's order
 * (sub-timestep, keyword, click) as it appends them to 'costs' / 'revenues_per_cost' (:97-104,113-115), and per keyword what
 * combine_outcomes (:124-147) leaves in 'impression_share' and 'profit' - its arithmetic literally, the lossy volume
 * re-derivation np.round(impressions / impression_share) included */
typedef struct {
    int64_t capacity, count;                       /* clicks that fit the lists / paid clicks of the step (may exceed capacity) */
    int32_t *env, *keyword, *timestep;             /* [capacity] */
    double *cost, *revenue;                        /* [capacity] dollars; revenue -1 = the click did not convert */
    double *impression_share, *profit;             /* [N][K] */
} orc_outcomes;

typedef struct {             /* persistent per-env state, caller-owned */
    float *params;           /* [P_COUNT][N][K] */
    uint64_t *key;           /* [N] */
    uint32_t *tick;          /* [N] */
    int32_t *day;            /* [N] */
    int64_t *cum_cents;      /* [N] IMPLICIT */
    double *cum;             /* [N] EXPLICIT */
    uint8_t *drift_pending;  /* [N] */
} orc_state;

static float P(const orc_state *s, const orc_config *c, int p, int env, int k)
{
    return s->params[((size_t)p * c->num_envs + env) * c->num_keywords + k];
}
static float *Pp(orc_state *s, const orc_config *c, int p, int env, int k)
{
    return &s->params[((size_t)p * c->num_envs + env) * c->num_keywords + k];
}

/* drift of one env's keywords (gymnasium_kw_env.py:132-158), applied lazily at the start of the
 * next step: values used by step t+1 equal the reference's values after step t's update. */
static void apply_drift(const orc_config *c, orc_state *s, int env, uint32_t tick_of_draw)
{
    for (int k = 0; k < c->num_keywords; ++k) {
        uint32_t w[4];
        draw(s->key[env], 0, ST_DRIFT, (uint32_t)k, tick_of_draw, w);
        float uv = fmaf(2.0f * c->drift_vol, u24(w[0]), -c->drift_vol);
        float uc = fmaf(2.0f * c->drift_ctr, u24(w[1]), -c->drift_ctr);
        float us = fmaf(2.0f * c->drift_cvr, u24(w[2]), -c->drift_cvr);
        float *vm = Pp(s, c, P_VOL_MEAN, env, k), *bc = Pp(s, c, P_BCTR, env, k), *sc = Pp(s, c, P_SCTR, env, k);
        float sd0 = P(s, c, P_VOL_STD, env, k);      /* :136-137 "init volume" is the vol std */
        *vm = fmaxf(fmaf(uv, sd0, *vm), 0.0f);       /* :146-149 */
        *bc = fminf(fmaxf(*bc * (1.0f + uc), 0.0f), 1.0f);   /* :153-155 */
        *sc = fminf(fmaxf(*sc * (1.0f + us), 0.0f), 1.0f);   /* :156-158 */
    }
}

/* the same update with the three coefficient vectors supplied by the caller (TAPE mode): the step calls
 * update_keywords() after the observation is formed (gymnasium_kw_env.py:246), so it is applied at once */
static void apply_drift_tape(const orc_config *c, orc_state *s, int env, const float *u)
{
    const size_t NK = (size_t)c->num_envs * c->num_keywords;
    for (int k = 0; k < c->num_keywords; ++k) {
        const size_t i = (size_t)env * c->num_keywords + k;
        const float uv = u[i], uc = u[NK + i], us = u[2 * NK + i];
        float *vm = Pp(s, c, P_VOL_MEAN, env, k), *bc = Pp(s, c, P_BCTR, env, k), *sc = Pp(s, c, P_SCTR, env, k);
        float sd0 = P(s, c, P_VOL_STD, env, k);      /* :136-137 */
        *vm = fmaxf(fmaf(uv, sd0, *vm), 0.0f);       /* :146-149 */
        *bc = fminf(fmaxf(*bc * (1.0f + uc), 0.0f), 1.0f);   /* :153-155 */
        *sc = fminf(fmaxf(*sc * (1.0f + us), 0.0f), 1.0f);   /* :156-158 */
    }
}

/* one visited cell's outcome merged into the keyword's running result: combine_outcomes, bidding_simulation.py:124-147 */
static void combine_cell(orc_outcomes *L, size_t idx, int32_t imps_before, int32_t cell_imps, int32_t cell_auctions, double cell_profit)
{
    /* simulate_epoch_of_bidding :88-91: the cell's own impression_share */
    const double cell_share = cell_auctions > 0 ? (double)cell_imps / (double)cell_auctions : 0.0;
    const double old_vol = imps_before < 1 ? 0.0 : rint((double)imps_before / L->impression_share[idx]);     /* :128-131 np.round */
    const double next_vol = cell_imps < 1 ? 0.0 : rint((double)cell_imps / cell_share);                      /* :132-135 */
    const double vol = old_vol + next_vol;                                                                   /* :141 */
    L->profit[idx] += cell_profit;                                                                           /* :136-137 */
    L->impression_share[idx] = vol > 0.0 ? (double)(imps_before + cell_imps) / vol : 0.0;                     /* :142-145 */
}

static void list_click(orc_outcomes *L, int env, int k, int t, double cost, double revenue)
{
    if (L->count < L->capacity) {
        const int64_t i = L->count;
        L->env[i] = env; L->keyword[i] = k; L->timestep[i] = t; L->cost[i] = cost; L->revenue[i] = revenue;
    }
    ++L->count;
}

static void step_env(const orc_config *c, orc_state *s, int env, const float *bids, float budget_in,
                     orc_tape *tape, orc_out *o, orc_outcomes *L)
{
    const int K = c->num_keywords;
    const size_t base = (size_t)env * K;
    const uint64_t key = s->key[env];
    const uint32_t tick = s->tick[env];
    const int use_tape = tape != NULL;

    if (!use_tape && c->drift_on && s->drift_pending[env]) {
        apply_drift(c, s, env, tick - 1u);
        s->drift_pending[env] = 0;
    }

    /* volumes and the 24-way split (bidding_simulation.py:151-167) */
    int32_t *V = o->volumes + base;
    for (int k = 0; k < K; ++k) {
        if (use_tape) V[k] = tape->volumes[base + k];
        else {
            uint32_t w[4];
            draw(key, 0, ST_VOL, (uint32_t)k >> 2, tick, w);        /* one call serves four consecutive keywords */
            V[k] = orc_volume_from_word(w[k & 3], P(s, c, P_VOL_MEAN, env, k), P(s, c, P_VOL_STD, env, k));
        }
        o->impressions[base + k] = o->clicks[base + k] = o->conversions[base + k] = 0;
        o->cost_cents[base + k] = o->revenue_cents[base + k] = 0;
        o->cost[base + k] = o->revenue[base + k] = 0.0;
        if (L) L->impression_share[base + k] = L->profit[base + k] = 0.0;
    }

    const int64_t budget_cents = orc_budget_cents(budget_in);
    double remaining_d = (double)budget_cents / 100.0;        /* np.round(budget, 2), gymnasium_kw_env.py:199 */
    double *profit_k = (double *)calloc((size_t)K, sizeof(double));   /* EXPLICIT per-keyword profit */
    int stop = 0;

    for (int t = 0; t < ORC_TIMESTEPS && !stop; ++t) {
        for (int k = 0; k < K && !stop; ++k) {
            const int32_t stepv = V[k] / ORC_TIMESTEPS;
            const int32_t n = t == 0 ? V[k] - (ORC_TIMESTEPS - 1) * stepv : stepv;
            const int32_t j0 = t == 0 ? 0 : V[k] - (ORC_TIMESTEPS - 1) * stepv + (t - 1) * stepv;
            const int64_t bid_c = orc_bid_cents(bids[base + k]);
            const uint64_t t_click = orc_bernoulli_threshold(P(s, c, P_BCTR, env, k));   /* EXPLICIT click words */
            const uint64_t t_conv = orc_bernoulli_threshold(P(s, c, P_SCTR, env, k));
            const float rev_mu = P(s, c, P_REV_MEAN, env, k), rev_sd = P(s, c, P_REV_STD, env, k);

            if (c->model == ORC_IMPLICIT) {
                /* simulate_epoch_of_bidding, bidding_simulation.py:86-117.  Money totals are carried in exact cents;
                 * the budget walk itself is done in binary floating point exactly as the reference does it (cost =
                 * cents/100 is the float np.around(x, 2) produced; `budget -= cost` per paid click; the campaign's
                 * remaining budget loses sum_list(costs), a left-to-right sum) because an exact tie - remaining ==
                 * cost - is decided there by the float residue, and whether the campaign then stops changes the
                 * impressions of every later cell. */
                const float loc = P(s, c, P_A, env, k), scale = P(s, c, P_B, env, k);
                double budget = remaining_d, cell_sum_d = 0.0, cell_rev_d = 0.0;
                int64_t cell_cost = 0;
                int32_t wins = 0, paid = 0, convs = 0;
                int broke = 0;
                /* pass 1: the auctions (impressions are not budget-limited, :86-88) */
                int64_t click_cur = tape ? tape->cur_click : 0;
                for (int32_t i = 0; i < n; ++i) {
                    int32_t click_bit = 0;
                    const uint32_t j = (uint32_t)(j0 + i);
                    int64_t comp;
                    uint32_t word = 0;
                    if (use_tape) comp = tape->bid_cents[tape->cur_bid++];
                    else {
                        uint32_t w[4];
                        draw(key, j >> 2, ST_AUCTION, (uint32_t)k, tick, w);
                        word = w[j & 3u];
                        comp = orc_auction_outcome(word, P(s, c, P_BCTR, env, k), loc, scale, &click_bit);
                    }
                    if (!(bid_c > comp)) continue;             /* tie loses, helpers.py:167-170 */
                    if (!use_tape && word == 0xFFFFFFFFu) continue;   /* stream rule: the word 2^32-1 never wins */
                    ++wins;
                    int clicked = use_tape ? tape->click[click_cur + wins - 1] : click_bit;
                    if (!clicked || broke) continue;
                    const double cost_d = (double)comp / 100.0;
                    if (budget >= cost_d) {                    /* :97-104 */
                        budget -= cost_d; cell_sum_d += cost_d; cell_cost += comp; ++paid;
                        uint32_t w2[4] = {0, 0, 0, 0};
                        if (!use_tape) draw(key, j, ST_CONV, (uint32_t)k, tick, w2);
                        int conv = use_tape ? tape->conv[tape->cur_conv++] : ((uint64_t)w2[0] < t_conv);
                        int64_t rev = 0;
                        if (conv) {
                            rev = use_tape ? tape->rev_cents[tape->cur_rev++]
                                           : orc_revenue_cents_tab(w2[1], rev_mu, rev_sd);
                            ++convs;
                            o->revenue_cents[base + k] += rev;
                            cell_rev_d += (double)rev / 100.0;       /* rust.sum_array(revenues), :117 */
                        }
                        if (L) list_click(L, env, k, t, cost_d, conv ? (double)rev / 100.0 : -1.0);
                    } else broke = 1;                          /* break: no later click of this cell is paid */
                }
                if (L) combine_cell(L, base + k, o->impressions[base + k], wins, n, cell_rev_d - cell_sum_d);
                if (use_tape) tape->cur_click += wins;
                o->impressions[base + k] += wins;
                o->clicks[base + k] += paid;
                o->conversions[base + k] += convs;
                o->cost_cents[base + k] += cell_cost;
                remaining_d -= cell_sum_d;                      /* bidding_simulation.py:225 (rust.sum_list: left to right) */
                if (remaining_d <= 0.0) stop = 1;               /* :230-233 */
            } else if (c->model == ORC_IMPLICIT_GENERAL) {
                /* the reference's default ImplicitKeyword (synthetic_kw_classes.py:610-686): B bidders for the whole call,
                 * raw Laplace bids, nth_price_auction(bid, other_bids, n=2, num_winners) (synthetic_kw_helpers.py:116-180)
                 * literally: top (w+n) bids per auction ascending (zero bids appended when there are fewer bidders),
                 * index = searchsorted left, won iff index > n, price = sorted[index - (n-1)].  Money is float64. */
                const float loc = P(s, c, P_A, env, k), scale = P(s, c, P_B, env, k);
                const double bid_d = (double)bid_c / 100.0;
                const int32_t top = c->num_winners + 2;
                double budget = remaining_d, cell_cost_sum = 0.0;
                int64_t cell_rev_c = 0;
                int32_t imps = 0, paid = 0, convs = 0;
                int broke = 0;
                int32_t B;
                if (use_tape) B = tape->x_impressions[tape->cur_ximp++];
                else {
                    float pmf0, ratio;
                    orc_bidder_law(c->max_bidders, c->participation_rate, &pmf0, &ratio);
                    if (pmf0 > 0.0f) {
                        uint32_t w[4];
                        draw(key, (uint32_t)(t >> 2), ST_GBIDDERS, (uint32_t)k, tick, w);
                        B = orc_bidders_from_word(w[t & 3], c->max_bidders, pmf0, ratio);
                    } else {
                        const uint64_t t_part = orc_bernoulli_threshold(c->participation_rate);
                        B = 0;
                        for (int32_t b = 0; b < c->max_bidders; ++b) {
                            uint32_t w[4];
                            draw(key, (uint32_t)(64 * t + (b >> 2)), ST_GBIDDERS, (uint32_t)k, tick, w);
                            if ((uint64_t)w[b & 3] < t_part) ++B;
                        }
                    }
                }
                const int32_t width = B > top ? B : top;
                double *row = (double *)malloc(sizeof(double) * (size_t)width);
                const int64_t bids_base = use_tape ? tape->cur_xcost : 0;
                int64_t click_cur = use_tape ? tape->cur_click : 0;
                for (int32_t i = 0; i < n; ++i) {
                    const uint32_t j = (uint32_t)(j0 + i);
                    int32_t m = 0;
                    for (int32_t z = 0; z < top - B; ++z) row[m++] = 0.0;            /* :156-161 zero bids */
                    if (use_tape) {
                        for (int32_t b = 0; b < B; ++b) row[m++] = tape->x_cost[bids_base + (int64_t)b * n + i];      /* (bidders, auctions) as drawn */
                    } else if (B > 0) {
                        /* the engine's stream (revision 3 for this model): only the top (w+n) bids of the B bidders matter below,
                         * and they are drawn directly as order statistics from one call (orc_top_laplace_bids) */
                        uint32_t w[4];
                        float xs[4];
                        draw(key, j, ST_GBID, (uint32_t)k, tick, w);
                        const int32_t nb = orc_top_laplace_bids(w, B, top, loc, scale, xs);
                        for (int32_t h = 0; h < nb; ++h) row[m++] = (double)xs[h];
                    }
                    qsort(row, (size_t)m, sizeof(double), cmp_f64);
                    const double *top_n = row + (m - top);                            /* the top (w+n), ascending (:152-155) */
                    int32_t index = 0;
                    while (index < top && top_n[index] < bid_d) ++index;              /* :167 searchsorted, side="left" */
                    if (!(index > 2)) continue;                                       /* :170 */
                    const double cost = top_n[index - 1];                             /* :173-175, n = 2 */
                    ++imps;
                    uint32_t w3[4] = {0, 0, 0, 0};
                    if (!use_tape) draw(key, j, ST_GCLICK, (uint32_t)k, tick, w3);
                    const int clicked = use_tape ? tape->click[click_cur + imps - 1] : ((uint64_t)w3[0] < t_click);
                    if (!clicked || broke) continue;
                    if (budget >= cost) {                                             /* bidding_simulation.py:97-104 */
                        budget -= cost; cell_cost_sum += cost; ++paid;
                        o->cost[base + k] += cost;
                        const int conv = use_tape ? tape->conv[tape->cur_conv++] : ((uint64_t)w3[1] < t_conv);
                        int64_t rev = 0;
                        if (conv) {
                            rev = use_tape ? tape->rev_cents[tape->cur_rev++] : orc_revenue_cents_tab(w3[2], rev_mu, rev_sd);
                            ++convs;
                            o->revenue_cents[base + k] += rev;
                            cell_rev_c += rev;
                        }
                        if (L) list_click(L, env, k, t, cost, conv ? (double)rev / 100.0 : -1.0);
                    } else broke = 1;
                }
                free(row);
                if (L) combine_cell(L, base + k, o->impressions[base + k], imps, n, (double)cell_rev_c / 100.0 - cell_cost_sum);
                if (use_tape) { tape->cur_xcost += (int64_t)B * n; tape->cur_click += imps; }
                o->impressions[base + k] += imps;
                o->clicks[base + k] += paid;
                o->conversions[base + k] += convs;
                profit_k[k] += (double)cell_rev_c / 100.0 - cell_cost_sum;
                remaining_d -= cell_cost_sum;
                if (remaining_d <= 0.0) stop = 1;
            } else {
                /* EXPLICIT cell: synthetic_kw_classes.py:535-538,514-518 then bidding_simulation.py:94-117 */
                const float bid_d = (float)((double)bid_c / 100.0);
                const float p_imp = orc_threshold_sigmoid_f32(bid_d, c->imp_thresh, P(s, c, P_A, env, k), P(s, c, P_B, env, k));
                const uint64_t t_imp = orc_bernoulli_threshold(p_imp);
                double budget = remaining_d, cell_cost_sum = 0.0;
                int64_t cell_rev_c = 0;        /* revenues are whole cents: summed exactly, converted once per cell */
                int32_t imps = 0, paid = 0, convs = 0;
                int broke = 0;
                int32_t n_imp_tape = use_tape ? tape->x_impressions[tape->cur_ximp++] : -1;
                int32_t loop_n = use_tape ? n_imp_tape : n;
                for (int32_t i = 0; i < loop_n; ++i) {
                    uint32_t w[4] = {0, 0, 0, 0};
                    double cost;
                    if (use_tape) cost = tape->x_cost[tape->cur_xcost++];
                    else {
                        draw(key, (uint32_t)(j0 + i), ST_AUCTION, (uint32_t)k, tick, w);
                        if (!((uint64_t)w[0] < t_imp)) continue;          /* Binomial(n,p) as n Bernoullis, src/lib.rs:70-76 */
                        cost = (double)orc_explicit_cost_from_word(w[1], bid_d);
                    }
                    ++imps;
                    int clicked = use_tape ? tape->click[tape->cur_click++] : ((uint64_t)w[2] < t_click);
                    if (!clicked || broke) continue;
                    if (budget >= cost) {
                        budget -= cost; cell_cost_sum += cost; ++paid;
                        o->cost[base + k] += cost;             /* obs cost = sum_list(costs): left-to-right */
                        int conv = use_tape ? tape->conv[tape->cur_conv++] : ((uint64_t)w[3] < t_conv);
                        int64_t rev = 0;
                        if (conv) {
                            if (use_tape) rev = tape->rev_cents[tape->cur_rev++];
                            else {
                                uint32_t w2[4];
                                draw(key, (uint32_t)(j0 + i), ST_XREV, (uint32_t)k, tick, w2);
                                rev = orc_revenue_cents_from_word(w2[0], rev_mu, rev_sd);
                            }
                            ++convs;
                            o->revenue_cents[base + k] += rev;
                            cell_rev_c += rev;
                        }
                        if (L) list_click(L, env, k, t, cost, conv ? (double)rev / 100.0 : -1.0);
                    } else broke = 1;
                }
                if (imps == 0) {
                    /* phantom: costs = np.array([0]) when impressions < 1 (synthetic_kw_classes.py:514-515),
                     * one zero-cost click opportunity per sub-timestep */
                    uint32_t w[4] = {0, 0, 0, 0};
                    if (!use_tape) draw(key, (uint32_t)t, ST_XPHANTOM, (uint32_t)k, tick, w);
                    int clicked = use_tape ? tape->click[tape->cur_click++] : ((uint64_t)w[0] < t_click);
                    if (clicked && budget >= 0.0) {
                        ++paid;
                        int conv = use_tape ? tape->conv[tape->cur_conv++] : ((uint64_t)w[1] < t_conv);
                        int64_t rev = 0;
                        if (conv) {
                            rev = use_tape ? tape->rev_cents[tape->cur_rev++]
                                           : orc_revenue_cents_from_word(w[2], rev_mu, rev_sd);
                            ++convs;
                            o->revenue_cents[base + k] += rev;
                            cell_rev_c += rev;
                        }
                        if (L) list_click(L, env, k, t, 0.0, conv ? (double)rev / 100.0 : -1.0);   /* the reference lists the zero cost */
                    }
                }
                if (L) combine_cell(L, base + k, o->impressions[base + k], imps, n, (double)cell_rev_c / 100.0 - cell_cost_sum);
                o->impressions[base + k] += imps;
                o->clicks[base + k] += paid;
                o->conversions[base + k] += convs;
                profit_k[k] += (double)cell_rev_c / 100.0 - cell_cost_sum;        /* combine_outcomes: profit += profit */
                remaining_d -= cell_cost_sum;
                if (remaining_d <= 0.0) stop = 1;
            }
        }
    }

    /* step tail, gymnasium_kw_env.py:222-244 */
    double reward;
    if (c->model == ORC_IMPLICIT) {
        int64_t profit_c = 0;
        for (int k = 0; k < K; ++k) {
            profit_c += o->revenue_cents[base + k] - o->cost_cents[base + k];
            o->cost[base + k] = (double)o->cost_cents[base + k] / 100.0;
            o->revenue[base + k] = (double)o->revenue_cents[base + k] / 100.0;
        }
        s->cum_cents[env] += profit_c;
        reward = (double)profit_c / 100.0;
        o->cum_profit[env] = (double)s->cum_cents[env] / 100.0;
    } else {
        reward = 0.0;
        for (int k = 0; k < K; ++k) {
            reward += profit_k[k];                              /* rust.sum_list, left to right */
            o->revenue[base + k] = (double)o->revenue_cents[base + k] / 100.0;
        }
        s->cum[env] += reward;
        o->cum_profit[env] = s->cum[env];
    }
    free(profit_k);
    o->reward[env] = reward;
    o->truncated[env] = o->cum_profit[env] < -c->loss_threshold;    /* :225 */
    s->day[env] += 1;                                               /* :227 */
    o->day[env] = s->day[env];
    o->terminated[env] = s->day[env] >= c->max_days;                /* :228 */
    s->tick[env] = tick + 1u;
    if (c->drift_on) s->drift_pending[env] = 1;                     /* :246 update_keywords() */
    if (use_tape && c->drift_on && tape->drift_uniforms) {          /* TAPE: the recorded coefficients, at once */
        apply_drift_tape(c, s, env, tape->drift_uniforms);
        s->drift_pending[env] = 0;
    }
    if (c->auto_reset && (o->terminated[env] || o->truncated[env])) {
        s->day[env] = 0; s->cum_cents[env] = 0; s->cum[env] = 0.0;  /* :327-328 */
    }
}

/* bids [N][K], budget [N]; tape (nullable) serves env 0..N-1 in order (cursors carry over); lists (nullable): the step's
 * combined outcomes click by click (serial over envs). */
ORC_API int32_t orc_step_outcomes(const orc_config *c, orc_state *s, const float *bids, const float *budget,
                                  orc_tape *tape, orc_out *o, orc_outcomes *lists)
{
    if (!c || !s || !bids || !budget || !o) return -1;
    if (c->num_envs <= 0 || c->num_keywords <= 0) return -1;
    if (!g_log_ready) build_log_table();
    if (!g_norm_ready) build_norm_table();
    if (lists) lists->count = 0;
    if (tape || lists || c->threads <= 1) {
        for (int e = 0; e < c->num_envs; ++e)
            step_env(c, s, e, bids, budget[e], tape, o, lists);
    } else {
#pragma omp parallel for schedule(dynamic, 4) num_threads(c->threads)
        for (int e = 0; e < c->num_envs; ++e)
            step_env(c, s, e, bids, budget[e], NULL, o, NULL);
    }
    return 0;
}

ORC_API int32_t orc_step(const orc_config *c, orc_state *s, const float *bids, const float *budget,
                         orc_tape *tape, orc_out *o)
{
    return orc_step_outcomes(c, s, bids, budget, tape, o, NULL);
}

/* force the pending drift into the stored parameters (what keyword_params shows after a step) */
ORC_API int32_t orc_materialize_drift(const orc_config *c, orc_state *s)
{
    for (int e = 0; e < c->num_envs; ++e)
        if (c->drift_on && s->drift_pending[e]) { apply_drift(c, s, e, s->tick[e] - 1u); s->drift_pending[e] = 0; }
    return 0;
}

/* synthetic action stream used by bench.py and the parity tests: bid = round2(U(lo, hi)) */
ORC_API void orc_sample_bids(const orc_config *c, const uint64_t *key, const uint32_t *tick, float lo, float hi, float *bids)
{
    for (int e = 0; e < c->num_envs; ++e)
        for (int k = 0; k < c->num_keywords; ++k) {
            uint32_t w[4];
            draw(key[e], 0, ST_ACTION, (uint32_t)k, tick[e], w);
            float b = fmaf(hi - lo, u24(w[0]), lo);
            bids[(size_t)e * c->num_keywords + k] = rintf(b * 100.0f) / 100.0f;
        }
}

/* ---- keyword-set generation, Philox form of gymnasium_kw_utils.py:295-339 + quantiles_to_keywords.py:13-28 ---- */
static float quantile_sample(const float *mins, const float *meds, const float *maxs, int32_t buckets, uint32_t wb, uint32_t wq)
{
    const uint32_t b = (uint32_t)(((uint64_t)wb * (uint32_t)buckets) >> 32);     /* rng.integers(0, B) */
    const float q = u24(wq);                                                       /* rng.random() */
    const float lo = mins[b], md = meds[b], hi = maxs[b];
    return q < 0.5f ? fmaf((md - lo) / 0.5f, q, lo) : fmaf((hi - md) / 0.5f, q - 0.5f, md);   /* np.interp(q, [0,.5,1], ...) */
}
/* tables: for quantity i (vol, ave_cpc, std_cpc, bctr, sctr, rpsc, std_rpsc) mins[i]/meds[i]/maxs[i] of length buckets[i];
 * params: [8][N][K] planes, written for every env */
ORC_API void orc_generate_implicit_keywords(int32_t N, int32_t K, const uint64_t *key, uint32_t serial, const int32_t *buckets,
                                            const float *const *mins, const float *const *meds, const float *const *maxs,
                                            float no_vol_prob, float *params)
{
    const uint32_t c3 = 0xFFFF0000u | (serial & 0xFFFFu);
    for (int e = 0; e < N; ++e)
        for (int k = 0; k < K; ++k) {
            uint32_t w[4][4];
            for (uint32_t i = 0; i < 4; ++i) {
                uint32_t c[4] = { i, ST_KEYGEN, (uint32_t)k, c3 }, kk[2] = { (uint32_t)key[e], (uint32_t)(key[e] >> 32) };
                orc_philox4x32(c, kk, w[i]);
            }
            float out[8];
            const float v = quantile_sample(mins[0], meds[0], maxs[0], buckets[0], w[0][0], w[0][1]);
            const float r = u24(w[0][3]);
            const int has = u24(w[0][2]) > no_vol_prob && v == v;          /* :298-300 */
            out[0] = has ? truncf(v) : 0.0f;
            out[1] = has ? truncf(fmaf(r * 0.5f, v, 1.0f)) : r * 0.5f;
            const float cpc = quantile_sample(mins[1], meds[1], maxs[1], buckets[1], w[1][0], w[1][1]);
            const float cpc_sd = quantile_sample(mins[2], meds[2], maxs[2], buckets[2], w[1][2], w[1][3]) * cpc;
            out[2] = cpc;
            out[3] = cpc_sd > 0.01f ? cpc_sd : 0.01f;                       /* :335-339 */
            out[4] = quantile_sample(mins[3], meds[3], maxs[3], buckets[3], w[2][0], w[2][1]);
            out[5] = quantile_sample(mins[4], meds[4], maxs[4], buckets[4], w[2][2], w[2][3]);
            const float rp = quantile_sample(mins[5], meds[5], maxs[5], buckets[5], w[3][0], w[3][1]);
            const float rp_sd = quantile_sample(mins[6], meds[6], maxs[6], buckets[6], w[3][2], w[3][3]) * rp;
            out[6] = rp;
            out[7] = rp_sd > 0.01f ? rp_sd : 0.01f;
            for (int p = 0; p < 8; ++p) params[((size_t)p * N + e) * K + k] = out[p];
        }
}

/* ---- keyword-set generation, EXPLICIT model: Philox form of sample_random_keywords (gymnasium_kw_utils.py:113-156) ----
 * Its Betas all have small integer parameters - (2,5), (5,2), (5,5) - and Beta(a, b) with integer a, b is the a-th smallest of
 * a + b - 1 independent uniforms.  Uniforms are the top 24 bits of Philox words, so every variate is i / 2^24. */
static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y ? 1 : 0;
}
static uint32_t order_statistic24(const uint32_t *w, int n, int kth)      /* the kth smallest (1-based) of w[0..n-1] >> 8 */
{
    uint32_t v[16];
    for (int i = 0; i < n; ++i) v[i] = w[i] >> 8;
    qsort(v, (size_t)n, sizeof(uint32_t), cmp_u32);
    return v[kth - 1];
}
/* int(2^x 15 - 1) for x = i / 2^24, :129-131, evaluated as the reference does (float64 pow; the device counts thresholds) */
ORC_API int32_t orc_explicit_vol_mean_from_i24(uint32_t i)
{
    return (int32_t)(pow(2.0, (double)i / 16777216.0) * 15.0 - 1.0);
}
/* params: [8][N][K] planes (vol_mean, vol_std, intercept, slope, bctr, sctr, rev_mean, rev_std), written for every env */
ORC_API void orc_generate_explicit_keywords(int32_t N, int32_t K, const uint64_t *key, uint32_t serial, float *params)
{
    const uint32_t c3 = 0xFFFF0000u | (serial & 0xFFFFu);
    const float s24 = 5.9604644775390625e-08f;
    for (int e = 0; e < N; ++e)
        for (int k = 0; k < K; ++k) {
            uint32_t w[52];
            for (uint32_t i = 0; i < 13; ++i) {
                uint32_t c[4] = { 16u + i, ST_KEYGEN, (uint32_t)k, c3 }, kk[2] = { (uint32_t)key[e], (uint32_t)(key[e] >> 32) };
                orc_philox4x32(c, kk, w + 4 * i);
            }
            float out[8];
            const int32_t vm = orc_explicit_vol_mean_from_i24(order_statistic24(w, 6, 2));       /* :129-131 beta(2, 5) */
            out[0] = (float)vm;
            out[1] = (u24(w[6]) * 0.5f) * (float)(vm + 1);                                      /* :133 */
            out[5] = (float)order_statistic24(w + 8, 6, 5) * s24;                               /* :135 beta(5, 2) */
            out[2] = u24(w[7]) * 1.5f;                                                          /* :136 */
            const float mu = ((float)order_statistic24(w + 16, 6, 2) * s24) * 1.5f;             /* :137 */
            out[6] = mu;
            out[7] = ((float)order_statistic24(w + 24, 6, 2) * s24) * mu;                       /* :138 */
            out[4] = (float)order_statistic24(w + 32, 6, 2) * s24;                              /* :139 */
            out[3] = ((float)order_statistic24(w + 40, 9, 5) * s24) * 25.0f;                    /* :140 beta(5, 5) */
            for (int p = 0; p < 8; ++p) params[((size_t)p * N + e) * K + k] = out[p];
        }
}

ORC_API int32_t orc_abi_version(void) { return 1; }
