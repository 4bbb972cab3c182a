// adc_shims.cpp - scalar entry points that mirror the reference's pyo3 module `adcraft.rust`
// function by function (src/lib.rs).  These are the host-side FFI a maintainer binds in place of
// the Rust crate; the per-step hot path does not go through them (it is adc_engine_step*).
//
// The reference's samplers draw from an unseeded thread_rng (src/lib.rs:25,43,61,75,320), so their
// individual values are not reproducible even by the reference; these take (seed, counter) and
// draw from the engine's Philox stream with the same float32 transforms the kernels use.
#include <cmath>
#include <cstdint>

#include "../../include/adcraft_engine.h"
#include "adc_law.h"

#define ADC_EXPORT extern "C" __attribute__((visibility("default")))

// src/lib.rs:290-294
ADC_EXPORT double adc_sigmoid(double x, double s, double t) { return 1.0 / (1.0 + std::exp(-s * (x - t))); }

// src/lib.rs:296-300 (num::clamp)
ADC_EXPORT double adc_clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// src/lib.rs:93-105
ADC_EXPORT double adc_threshold_sigmoid(double p, double impression_thresh, double impression_bid_intercept,
                                        double impression_slope)
{
    const double halver = 2.0 + 1e-10;
    const double thresh = adc_clamp(halver * impression_thresh, 0.0, 1.0) / halver;
    const double r = adc_sigmoid(p, impression_slope, impression_bid_intercept);
    return adc_clamp((1.0 + 2.0 * thresh) * r - thresh, 0.0, 1.0);
}

// src/lib.rs:108-116,310-312: sequential left-to-right f64 sum
ADC_EXPORT double adc_sum_f64(const double *x, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += x[i];
    return s;
}

// src/lib.rs:119-127
ADC_EXPORT int64_t adc_count_true(const uint8_t *x, int64_t n)
{
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) c += x[i] != 0;
    return c;
}

static inline adc::U4 shim_draw(uint64_t seed, uint64_t counter, uint32_t lane)
{
    return adc::philox4x32((uint32_t)counter, (uint32_t)(counter >> 32), lane, 0x5348494Du /* "SHIM" */,
                              (uint32_t)seed, (uint32_t)(seed >> 32));
}

// src/lib.rs:314-325: round(max(N(mean, std), 0)), f64::round = half away from zero
ADC_EXPORT uint64_t adc_nonneg_int_normal(double mean, double std, uint64_t seed, uint64_t counter)
{
    const adc::U4 w = shim_draw(seed, counter, 0);
    double x = mean + std * (double)adc::normal_from_word(w.x);
    if (!(x > 0.0)) x = 0.0;
    return (uint64_t)std::round(x);
}

// src/lib.rs:70-76: Binomial(n, p) as n Bernoulli draws
ADC_EXPORT uint64_t adc_binomial(uint64_t n, double p, uint64_t seed, uint64_t counter)
{
    const uint64_t thr = adc::bernoulli_threshold((float)p);
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; i += 4) {
        const adc::U4 w = shim_draw(seed, counter, (uint32_t)(1 + i / 4));
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        for (uint64_t j = 0; j < 4 && i + j < n; ++j) c += adc::bernoulli(ws[j], thr);
    }
    return c;
}

// src/lib.rs:54-67: clamp(sqrt(x)/4 + 4.4/2 + N(0, 1e-10 + sqrt(x)/6), 0, 4.4) - note the constant 4.4
ADC_EXPORT int adc_cost_create(double x, int64_t n, uint64_t seed, uint64_t counter, double *out)
{
    if (n < 0 || (n > 0 && !out) || !(x >= 0.0)) return ADC_EINVAL;
    for (int64_t i = 0; i < n; i += 4) {
        const adc::U4 w = shim_draw(seed, counter, (uint32_t)(1 + i / 4));
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        for (int64_t j = 0; j < 4 && i + j < n; ++j) out[i + j] = (double)adc::explicit_cost(ws[j], (float)x);
    }
    return ADC_OK;
}

// The word-space form of the 2nd-price clearing used by k_step_implicit_fast, evaluated on the host (same code,
// adc_law.h): for a keyword (bid, competitor law, click rate) the auction word w is a clicked win iff
// (w - out[0]) < out[1] and an unclicked win iff (w - out[2]) < out[3] in unsigned 32-bit arithmetic.  Diagnostic entry
// point: lets a caller (and tests/test_abi_and_host.py) check the thresholds against auction-by-auction resolution
// (adcraft/synthetic_kw_helpers.py:116-180 on the sampled competitor bid) without a GPU.
static const adc::LogTableEntry *host_log_table()
{
    static adc::LogTableEntry table[adc::kLogTableIntervals];
    static bool ready = false;
    if (!ready) {
        for (int i = 0; i < adc::kLogTableIntervals; ++i) table[i] = adc::log_table_entry(i);
        ready = true;
    }
    return table;
}

ADC_EXPORT int adc_auction_word_intervals(float bid, float cost_loc, float cost_scale, float buyside_ctr, uint32_t *out4)
{
    if (!out4) return ADC_EINVAL;
    const adc::LogTableEntry *table = host_log_table();
    const adc::AuctionLaw law = adc::make_auction_law(buyside_ctr);
    const adc::WinIntervals r = adc::win_intervals((int32_t)adc::bid_to_cents(bid), cost_loc, cost_scale,
                                                   adc::bernoulli_threshold(buyside_ctr), law, table);
    out4[0] = r.c_lo; out4[1] = r.c_w; out4[2] = r.n_lo; out4[3] = r.n_w;
    return ADC_OK;
}

// The conservative brackets of those two intervals that k_step_implicit_sparse classifies auctions with (adc_law.h
// win_brackets): out8 = {outer c_lo, c_w, n_lo, n_w, inner c_lo, c_w, n_lo, n_w}.  Host evaluation of the same code (the
// device's exp2 / rcp differ in the last bits; both stay inside the slack - adc_engine_debug_win_brackets is the device's).
ADC_EXPORT int adc_auction_word_brackets(float bid, float cost_loc, float cost_scale, float buyside_ctr, uint32_t *out8)
{
    if (!out8) return ADC_EINVAL;
    const adc::WinBrackets b = adc::win_brackets((int32_t)adc::bid_to_cents(bid), cost_loc, cost_scale, adc::bernoulli_threshold_f32(buyside_ctr));
    out8[0] = b.out.c_lo; out8[1] = b.out.c_w; out8[2] = b.out.n_lo; out8[3] = b.out.n_w;
    out8[4] = b.in.c_lo; out8[5] = b.in.c_w; out8[6] = b.in.n_lo; out8[7] = b.in.n_w;
    return ADC_OK;
}

// [lo, lo + w) as a pair of 64-bit ends; an empty interval has w == 0
static inline bool interval_inside(uint32_t a_lo, uint32_t a_w, uint32_t b_lo, uint32_t b_w)
{
    if (a_w == 0u) return true;
    return b_w != 0u && a_lo >= b_lo && (uint64_t)a_lo + a_w <= (uint64_t)b_lo + b_w;
}

// Checks inner subset-of exact subset-of outer for n keywords; brackets8 == NULL: the host's own win_brackets, otherwise the given
// ones (8 words per keyword, e.g. computed on the device).  Returns the number of keywords that violate an inclusion
// (first_bad = index of the first, or -1); *ambiguous_words (optional) = total width of the in-between zones, the
// long-way share of the stream.
ADC_EXPORT int64_t adc_check_win_brackets(int64_t n, const float *bid, const float *cost_loc, const float *cost_scale, const float *buyside_ctr,
                                          const uint32_t *brackets8, int64_t *first_bad, double *ambiguous_words)
{
    if (n < 0 || !bid || !cost_loc || !cost_scale || !buyside_ctr) return -1;
    const adc::LogTableEntry *table = host_log_table();
    int64_t bad = 0, first = -1;
    double amb = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t bid_c = (int32_t)adc::bid_to_cents(bid[i]);
        const adc::AuctionLaw law = adc::make_auction_law(buyside_ctr[i]);
        const adc::WinIntervals x = adc::win_intervals(bid_c, cost_loc[i], cost_scale[i], adc::bernoulli_threshold(buyside_ctr[i]), law, table);
        adc::WinBrackets b;
        if (brackets8) {
            const uint32_t *p = brackets8 + 8 * i;
            b.out = adc::WinIntervals{p[0], p[1], p[2], p[3]};
            b.in = adc::WinIntervals{p[4], p[5], p[6], p[7]};
        } else {
            b = adc::win_brackets(bid_c, cost_loc[i], cost_scale[i], adc::bernoulli_threshold_f32(buyside_ctr[i]));
        }
        const bool ok = interval_inside(b.in.c_lo, b.in.c_w, x.c_lo, x.c_w) && interval_inside(x.c_lo, x.c_w, b.out.c_lo, b.out.c_w) &&
                        interval_inside(b.in.n_lo, b.in.n_w, x.n_lo, x.n_w) && interval_inside(x.n_lo, x.n_w, b.out.n_lo, b.out.n_w) &&
                        (uint64_t)b.out.c_lo + b.out.c_w <= 0xFFFFFFFFull && (uint64_t)b.out.n_lo + b.out.n_w <= 0xFFFFFFFFull;
        if (!ok) { if (first < 0) first = i; ++bad; }
        amb += ((double)b.out.c_w - (double)b.in.c_w) + ((double)b.out.n_w - (double)b.in.n_w);
    }
    if (first_bad) *first_bad = first;
    if (ambiguous_words) *ambiguous_words = amb;
    return bad;
}

// adcraft/gymnasium_kw_utils.py:113-156 (sample_random_keywords), one keyword: what k_generate_explicit_keywords writes
ADC_EXPORT int adc_sample_random_keyword(uint64_t key, uint32_t keyword, uint32_t serial, float *out8)
{
    if (!out8) return ADC_EINVAL;
    adc::generate_explicit_keyword(key, keyword, serial, out8);
    return ADC_OK;
}
