// adc_law.h - the engine's random stream and sampling law (gfx950 device code; the few scalar
// FFI shims reuse it on the host).
//
// Stream (revision 2): Philox4x32-7 (Salmon et al., SC11; 7 rounds is the Crush-resistant variant of the paper's Table 2,
// the 10-round default adds a safety margin), counter-addressed, no per-lane state in HBM:
//     words = philox(key = env key (64 bit), ctr = (index, stage, keyword, tick))
// so a variate depends only on WHAT it is for - never on which lane, wave, block or GPU drew it.
// Transforms are float32 and use only operations that IEEE-754 rounds correctly
// (+ - * / sqrt fma rint), each written out explicitly (build with -ffp-contract=off), which is
// what lets a CPU restatement reproduce every integer this file produces.
//
// What each transform stands for in the reference:
//   competitor_cents_from_v   round2(max(|Laplace(loc,scale)|, 0))   adcraft/synthetic_kw_helpers.py:104-113
//   bernoulli        rng.random(n) <= p                     adcraft/synthetic_kw_helpers.py:73-77
//   revenue_cents    round2(max(N(mu,sd), 0.01))            adcraft/synthetic_kw_helpers.py:66-70
//   volume           round(max(N(mean,sd), 0))              src/lib.rs:314-325
//   explicit_cost    clamp(sqrt(x)/4 + 2.2 + N(0,.), 0, 4.4) src/lib.rs:54-67
//   threshold_sigmoid                                        src/lib.rs:93-105
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ADC_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define ADC_HD static inline
#endif

namespace adc {

// both:     call (0, ST_VOL, k/4)  = the volume words of keywords 4*(k/4) .. +3 (x,y,z,w for k%4 = 0..3)
// IMPLICIT: call (j/4, ST_AUCTION) = one word per auction j (x,y,z,w for j%4 = 0..3); the word decides the
//           click AND supplies the competitor-bid uniform (interval splitting, see AuctionLaw);
//           call (j, ST_CONV)      = {conversion, revenue} words (x,y), consumed only for a paid click.
// EXPLICIT: call (j, ST_AUCTION)   = {impression, cost, click, conversion}; (j, ST_XREV).x = revenue;
//           (t, ST_XPHANTOM)       = {click, conversion, revenue} of the zero-impression phantom of cell t.
// GENERAL (the reference's default ImplicitKeyword; stream revisions 3 and 4 for this model): call (t/4, ST_GBIDDERS) word t%4 =
//           the uniform that the bidder count of sub-timestep t is read off (general_bidder_law: Binomial by inversion); where
//           that is not applicable, call (64 t + b/4, ST_GBIDDERS) word b%4 = participation coin of bidder b, as before;
//           call (j, ST_GBID) = the exponential spacings of the TOP bids of
//           auction j, highest first (top_laplace_bids: order statistics instead of one draw per bidder);
//           call (j, ST_GCLICK) = {click, conversion, revenue} words of auction j.
enum Stage : uint32_t { ST_VOL = 0, ST_AUCTION = 1, ST_DRIFT = 2, ST_XPHANTOM = 3, ST_XREV = 4, ST_ACTION = 5, ST_METRIC = 6, ST_CONV = 7, ST_KEYGEN = 8, ST_AGENT = 9,
                        ST_GBIDDERS = 10, ST_GBID = 11, ST_GCLICK = 12 };
constexpr int kTimesteps = 24;          // adcraft/bidding_simulation.py:213
constexpr int kVolumeMax = 1 << 20;
constexpr float kMoneyMaxCents = 1.0e9f;

struct U4 { uint32_t x, y, z, w; };

ADC_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);      // one VALU op on gfx950
#else
    return a ^ b ^ c;
#endif
}

ADC_HD void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1)
{
    const uint64_t a = (uint64_t)0xD2511F53u * c0;          // v_mad_u64_u32: both halves in one op
    const uint64_t b = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = xor3((uint32_t)(b >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(a >> 32), c3, k1);
    c1 = (uint32_t)b;
    c3 = (uint32_t)a;
    c0 = n0;
    c2 = n2;
}

constexpr int kPhiloxRounds = 7;

ADC_HD U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < kPhiloxRounds; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

ADC_HD U4 draw(uint64_t key, uint32_t index, uint32_t stage, uint32_t keyword, uint32_t tick)
{
    return philox4x32(index, stage, keyword, tick, (uint32_t)key, (uint32_t)(key >> 32));
}

// the volume word of keyword k: one call serves four consecutive keywords
ADC_HD uint32_t volume_word(uint64_t key, uint32_t k, uint32_t tick)
{
    const U4 w = draw(key, 0u, 0u /* ST_VOL */, k >> 2, tick);
    const uint32_t h = k & 3u;
    return h == 0u ? w.x : h == 1u ? w.y : h == 2u ? w.z : w.w;
}

// ---- bit casts ---------------------------------------------------------------------------------
ADC_HD float bits_to_float(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { uint32_t u; float f; } v; v.u = u; return v.f;
#endif
}
ADC_HD uint32_t float_to_bits(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    union { uint32_t u; float f; } v; v.f = f; return v.u;
#endif
}
ADC_HD float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- deterministic log / exp (Cephes single-precision schemes) ----------------------------------
ADC_HD float det_log(float x)
{
    const uint32_t u = float_to_bits(x);
    int e = (int)(u >> 23) - 126;
    float m = bits_to_float((u & 0x007FFFFFu) | 0x3F000000u);       // [0.5, 1)
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292E-2f;
    p = fma32(p, f, -1.1514610310E-1f);
    p = fma32(p, f, 1.1676998740E-1f);
    p = fma32(p, f, -1.2420140846E-1f);
    p = fma32(p, f, 1.4249322787E-1f);
    p = fma32(p, f, -1.6668057665E-1f);
    p = fma32(p, f, 2.0000714765E-1f);
    p = fma32(p, f, -2.4999993993E-1f);
    p = fma32(p, f, 3.3333331174E-1f);
    float y = f * (z * p);
    const float fe = (float)e;
    y = fma32(fe, -2.12194440e-4f, y);
    y = fma32(z, -0.5f, y);
    float r = f + y;
    r = fma32(fe, 0.693359375f, r);
    return r;
}

ADC_HD float det_exp(float x)
{
    x = x > 87.0f ? 87.0f : x;
    x = x < -87.0f ? -87.0f : x;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = fma32(n, -0.693359375f, x);
    r = fma32(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500E-4f;
    p = fma32(p, r, 1.3981999507E-3f);
    p = fma32(p, r, 8.3334519073E-3f);
    p = fma32(p, r, 4.1665795894E-2f);
    p = fma32(p, r, 1.6666665459E-1f);
    p = fma32(p, r, 5.0000001201E-1f);
    const float y = fma32(p, z, r) + 1.0f;
    return y * bits_to_float((uint32_t)((int)n + 127) << 23);
}

// ---- uniforms -----------------------------------------------------------------------------------
ADC_HD float unit_open23(uint32_t w) { return ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-07f; }   // (0,1)
ADC_HD float unit_half24(uint32_t w) { return ((float)(w >> 9) + 0.5f) * 5.9604644775390625e-08f; }   // (0,0.5)
ADC_HD float unit_closed24(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }          // [0,1)
ADC_HD bool sign_bit(uint32_t w) { return (w >> 8) & 1u; }

// standard normal by inversion, Wichura AS241 PPND7; lower-tail probability from bits 31..9, sign = bit 8
ADC_HD float normal_from_word(uint32_t w)
{
    const float p = unit_half24(w);
    const float q = p - 0.5f;
    float val;
    if (q >= -0.425f) {
        const float r = fma32(-q, q, 0.180625f);
        const float num = fma32(fma32(fma32(5.9109374720e1f, r, 1.5929113202e2f), r, 5.0434271938e1f), r, 3.3871327179e0f);
        const float den = fma32(fma32(fma32(6.7187563600e1f, r, 7.8757757664e1f), r, 1.7895169469e1f), r, 1.0f);
        val = q * num / den;
    } else {
        const float r = __builtin_sqrtf(-det_log(p)) - 1.6f;
        const float num = fma32(fma32(fma32(1.7023821103e-1f, r, 1.3067284816e0f), r, 2.7568153900e0f), r, 1.4234372777e0f);
        const float den = fma32(fma32(1.2021132975e-1f, r, 7.3700164250e-1f), r, 1.0f);
        val = -(num / den);
    }
    return sign_bit(w) ? -val : val;
}

// ---- standard normal by table (revision 2) --------------------------------------------------------
// One word -> N(0,1) by inversion: bit 31 = sign, bits 30..8 = m, t = 2m + 1 (odd, 24 bits), lower-tail probability
// p = t / 2^25 in (0, 1/2).  |z| = -Phi^-1(p) is read from a table indexed by the FLOAT representation of t: 24 octaves x
// 32 intervals (the float exponent and the top 5 mantissa bits), linear in the remaining 18 mantissa bits.  The nodes are
// AS241 PPND7 values (the deterministic float32 code above), so host and device build the same bits; the chord error is
// below 6e-5 in z everywhere (convex function: 32 intervals per octave), i.e. < 0.01 cent on a typical revenue.
struct NormTableEntry { float value, slope; };
constexpr int kNormTableEntries = 24 * 32;

// -Phi^-1(p) >= 0 for p in (0, 1/2]: AS241 PPND7, central and intermediate branches (p >= 2^-25 keeps r below 5)
ADC_HD float normal_tail_magnitude(float p)
{
    const float q = p - 0.5f;
    if (q >= -0.425f) {
        const float r = fma32(-q, q, 0.180625f);
        const float num = fma32(fma32(fma32(5.9109374720e1f, r, 1.5929113202e2f), r, 5.0434271938e1f), r, 3.3871327179e0f);
        const float den = fma32(fma32(fma32(6.7187563600e1f, r, 7.8757757664e1f), r, 1.7895169469e1f), r, 1.0f);
        return -(q * num / den);
    }
    const float r = __builtin_sqrtf(-det_log(p)) - 1.6f;
    const float num = fma32(fma32(fma32(1.7023821103e-1f, r, 1.3067284816e0f), r, 2.7568153900e0f), r, 1.4234372777e0f);
    const float den = fma32(fma32(1.2021132975e-1f, r, 7.3700164250e-1f), r, 1.0f);
    return num / den;
}

ADC_HD float norm_table_node(int i)            // |z| at the left end of interval i; node 768 is p = 1/2
{
    if (i >= kNormTableEntries) return 0.0f;
    const float t0 = bits_to_float((uint32_t)(127 + (i >> 5)) << 23) * (1.0f + (float)(i & 31) * 0.03125f);     // 2^e (1 + m/32)
    return normal_tail_magnitude(t0 * 2.98023223876953125e-08f);                                                  // p = t0 / 2^25
}

ADC_HD NormTableEntry norm_table_entry(int i)
{
    const float a = norm_table_node(i), b = norm_table_node(i + 1);
    return NormTableEntry{a, (b - a) * 3.814697265625e-06f};            // slope per unit of the 18 low mantissa bits
}

ADC_HD float normal_tab(uint32_t w, const NormTableEntry *tab)
{
    const uint32_t t = ((w >> 7) & 0x00FFFFFEu) | 1u;                   // 2 * bits(30..8) + 1
    const uint32_t bits = float_to_bits((float)t);                      // exact: t < 2^24
    const NormTableEntry e = tab[(bits >> 18) - (127u << 5)];
    const float z = fma32(e.slope, (float)(bits & 0x0003FFFFu), e.value);
    return bits_to_float(float_to_bits(z) | (~w & 0x80000000u));        // sign bit set in the word: +|z|, clear: -|z|
}

ADC_HD int32_t money_to_cents(float dollars)
{
    float c = __builtin_rintf(dollars * 100.0f);
    c = c < kMoneyMaxCents ? c : kMoneyMaxCents;
    return (int32_t)c;
}

// event <=> (uint64)word < threshold; threshold = round(p * 2^32) in [0, 2^32]
// (float32 throughout, exactly: p 2^32 is a power-of-two scaling; at or above 2^23 it is already an integer, below it
// x + 0.5 is representable - so this is floor(p 2^32 + 0.5) as a double would compute it)
ADC_HD float bernoulli_threshold_f32(float p)
{
    if (!(p > 0.0f)) return 0.0f;                        // (NaN too)
    const float x = p * 4294967296.0f;
    if (!(x < 4294967296.0f)) return 4294967296.0f;
    return x >= 8388608.0f ? x : __builtin_floorf(x + 0.5f);
}
ADC_HD uint64_t bernoulli_threshold(float p)
{
    const float t = bernoulli_threshold_f32(p);
    return t >= 4294967296.0f ? 4294967296ull : (uint64_t)(uint32_t)t;
}
ADC_HD bool bernoulli(uint32_t w, uint64_t threshold) { return (uint64_t)w < threshold; }
// the same test with the threshold saturated to 32 bits: 0xFFFFFFFF can only come from T = 2^32 (p = 1; the
// largest float32 below 1 gives T = 2^32 - 256), so it encodes "always"
ADC_HD uint32_t saturate_threshold(uint64_t t) { return t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t; }
ADC_HD bool bernoulli32(uint32_t w, uint32_t t32) { return (w < t32) | (t32 == 0xFFFFFFFFu); }

// -log(u) for u = w24 / 2^24, w24 odd in [1, 2^24): exponent from the float32 bits, mantissa through a
// 256-interval table of det_log on [1,2) with linear interpolation (error < 4e-6, i.e. < 4e-5 cent of a typical
// bid - far below the cent rounding).  tab[i] = {det_log(1 + i/256), (det_log(1 + (i+1)/256) - det_log(1 + i/256)) * 2^-15}: the table
// itself is built with det_log, so host and device hold the same bits.
struct LogTableEntry { float value, slope; };
constexpr int kLogTableIntervals = 256;

ADC_HD LogTableEntry log_table_entry(int i)
{
    const float a = det_log(1.0f + (float)i * 0.00390625f);
    const float b = det_log(1.0f + (float)(i + 1) * 0.00390625f);
    return LogTableEntry{a, (b - a) * 3.0517578125e-05f};
}

// the table as its node values alone (kLogTableIntervals + 1 floats: half the bytes, for LDS copies): the slope is rebuilt
// from two neighbouring nodes with the operations log_table_entry used, so both forms give the same bits
struct LogNodes { const float *nodes; };
ADC_HD LogTableEntry log_entry(const LogTableEntry *tab, uint32_t i) { return tab[i]; }
ADC_HD LogTableEntry log_entry(LogNodes t, uint32_t i)
{
    const float a = t.nodes[i], b = t.nodes[i + 1];
    return LogTableEntry{a, (b - a) * 3.0517578125e-05f};
}
ADC_HD float log_table_node(int i) { return det_log(1.0f + (float)i * 0.00390625f); }      // i = 0 .. kLogTableIntervals

ADC_HD float neg_log_u24(uint32_t w24, const LogTableEntry *tab)
{
    const uint32_t bits = float_to_bits((float)w24);                 // exact: w24 < 2^24
    const float ef = (float)((int)(bits >> 23) - 151);              // exponent of u = w24 * 2^-24, in [-24, -1]
    const uint32_t mant = bits & 0x007FFFFFu;
    const LogTableEntry t = tab[mant >> 15];
    float r = fma32(t.slope, (float)(mant & 0x7FFFu), t.value);
    r = fma32(ef, -2.12194440e-4f, r);
    r = fma32(ef, 0.693359375f, r);
    return -r;
}

// One word per auction.  click <=> word < T with T = round(ctr * 2^32) (as every Bernoulli here).  Inside
// either outcome the word is still uniform on its sub-interval [0,T) or [T,2^32), so rescaling its offset d
// to 24 bits, i24 = floor(d * 2^24 / range) computed as mulhi(d, floor(2^56 / range)), gives a uniform that is
// independent of the click: bit 0 = sign, bits 23..1 = magnitude of the Laplace competitor bid.  (Integer
// rescaling on purpose: a float32 product cannot carry 32 bits and correlates the sign bit with the magnitude.)
struct AuctionLaw {
    uint32_t t32;          // click threshold T saturated to 32 bits (0xFFFFFFFF <=> T = 2^32: every word clicks)
    uint32_t m_click;      // floor(2^56 / T)            (0 if T == 0)
    uint32_t m_noclick;    // floor(2^56 / (2^32 - T))   (0 if T == 2^32)
};

// floor(2^56 / range) in float32 (one correctly rounded division; 24 good bits are plenty: the map d -> i24
// stays monotone and each i24 value keeps its share of words to within one word), saturated to 32 bits
ADC_HD uint32_t rescale_multiplier(uint64_t range)
{
    if (range == 0) return 0u;
    float m = __builtin_floorf(72057594037927936.0f / (float)range);
    m = m < 4294967040.0f ? m : 4294967040.0f;
    return (uint32_t)m;
}

// rescale_multiplier of a range given as the float32 it converts to
ADC_HD uint32_t rescale_multiplier_f32(float range)
{
    if (!(range > 0.0f)) return 0u;
    float m = __builtin_floorf(72057594037927936.0f / range);
    m = m < 4294967040.0f ? m : 4294967040.0f;
    return (uint32_t)m;
}

ADC_HD AuctionLaw make_auction_law(float bctr)
{
    // T is an integer-valued float32 <= 2^32, so (float)T is T itself and (float)(2^32 - T) is the float32 difference
    // (one rounding of the same integer either way): no 64-bit integer conversions
    const float t = bernoulli_threshold_f32(bctr);
    AuctionLaw a;
    a.t32 = t >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)t;
    a.m_click = rescale_multiplier_f32(t);
    a.m_noclick = rescale_multiplier_f32(4294967296.0f - t);
    return a;
}

ADC_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// The 24-bit uniform v of an auction, laid out so that the competitor's bid is MONOTONE in it (revision 2): bit 23 = sign
// of the Laplace deviate z, bits 22..0 = magnitude index, reflected on the positive side:
//     v < 2^23 :  z = -e(mag),  mag = v            (v up  =>  u up  =>  e = -log u down  =>  z up)
//     v >= 2^23:  z = +e(mag),  mag = 2^24 - 1 - v (v up  =>  mag down  =>  e up  =>  z up)
// with u = (2 mag + 1) / 2^24 and e(mag) = -log u by the table above.  X = loc + |scale| z and round(100 X) are then
// non-decreasing in v (every step is a correctly rounded monotone operation), so {v : bid > cents(v)} is an interval - which
// is what lets the step kernels resolve an auction by comparing its word with per-keyword thresholds.
// neg_log_u24 with its argument already a float (an odd integer below 2^24, exactly representable)
template <typename Tab>
ADC_HD float neg_log_f24(float w24, Tab tab)
{
    const uint32_t bits = float_to_bits(w24);
    const float ef = (float)((int)(bits >> 23) - 151);
    const uint32_t mant = bits & 0x007FFFFFu;
    const LogTableEntry t = log_entry(tab, mant >> 15);
    float r = fma32(t.slope, (float)(mant & 0x7FFFu), t.value);
    r = fma32(ef, -2.12194440e-4f, r);
    r = fma32(ef, 0.693359375f, r);
    return -r;
}

template <typename Tab>
ADC_HD float laplace_deviate_from_v(uint32_t v24, Tab tab)
{
    const uint32_t neg_mask = 0u - (v24 >> 23);                                  // all ones on the positive side
    const uint32_t mag = (v24 ^ neg_mask) & 0x007FFFFFu;
    const float e = neg_log_f24(fma32((float)mag, 2.0f, 1.0f), tab);            // 2 mag + 1 < 2^24: exact in float32; e > 0
    return bits_to_float(float_to_bits(e) ^ (~neg_mask & 0x80000000u));         // negative side: -e
}

// round(100 X) with its sign, clamped to +-1e9: non-decreasing in v.  |.| of it is the competitor's bid in cents.
template <typename Tab>
ADC_HD int32_t signed_cents_from_v(uint32_t v24, float loc, float scale, Tab tab)
{
    const float x = fma32(__builtin_fabsf(scale), laplace_deviate_from_v(v24, tab), loc);
    float c = __builtin_rintf(x * 100.0f);
    c = c < kMoneyMaxCents ? c : kMoneyMaxCents;          // (NaN -> +1e9: a NaN parameter never wins)
    c = c > -kMoneyMaxCents ? c : -kMoneyMaxCents;
    return (int32_t)c;
}

// |signed_cents_from_v|: rint and the clamp are odd-symmetric, so |round(100 x)| = round(100 |x|)
template <typename Tab>
ADC_HD int32_t competitor_cents_from_v(uint32_t v24, float loc, float scale, Tab tab)
{
    const float x = fma32(__builtin_fabsf(scale), laplace_deviate_from_v(v24, tab), loc);
    float c = __builtin_rintf(__builtin_fabsf(x) * 100.0f);
    c = c < kMoneyMaxCents ? c : kMoneyMaxCents;          // (NaN -> 1e9: a NaN parameter never wins)
    return (int32_t)c;
}

// the 24-bit uniform of an auction word (click decided, offset rescaled inside its sub-interval)
ADC_HD uint32_t auction_uniform24(uint32_t w, const AuctionLaw &a, bool &click)
{
    click = bernoulli32(w, a.t32);
    const uint32_t d = click ? w : w - a.t32;
    const uint32_t v = mulhi32(d, click ? a.m_click : a.m_noclick);
    return v < 0x00FFFFFFu ? v : 0x00FFFFFFu;
}

// 2nd-price clearing against one competitor: the bid must exceed the competitor's (a tie loses,
// adcraft/synthetic_kw_helpers.py:167-170).  The single word 2^32 - 1 never wins (revision 2): it keeps every threshold of
// the word-space form of this test (k_step_implicit_fast) within 32 bits, at a cost of 2^-32 in win probability.
ADC_HD bool auction_wins(uint32_t w, int32_t bid_c, int32_t comp_c) { return bid_c > comp_c && w != 0xFFFFFFFFu; }

// auction_outcome for callers that apply auction_wins right away: the click of the word 2^32 - 1 (the only word the
// saturated threshold 0xFFFFFFFF = "always" is needed for) is immaterial because that word never wins
template <typename Tab>
ADC_HD int32_t auction_outcome_unless_top_word(uint32_t w, const AuctionLaw &a, float loc, float scale, Tab tab, bool &click)
{
    click = w < a.t32;
    const uint32_t d = click ? w : w - a.t32;
    const uint32_t v = mulhi32(d, click ? a.m_click : a.m_noclick);
    return competitor_cents_from_v(v < 0x00FFFFFFu ? v : 0x00FFFFFFu, loc, scale, tab);
}

ADC_HD int32_t auction_outcome(uint32_t w, const AuctionLaw &a, float loc, float scale, const LogTableEntry *tab, bool &click)
{
    return competitor_cents_from_v(auction_uniform24(w, a, click), loc, scale, tab);
}

// ---- the auction resolved in word space --------------------------------------------------------------------------
// signed_cents_from_v is non-decreasing in v, so "the bid exceeds the competitor's" is v in [W_lo, W_hi) with
//     W_lo = min{v : S(v) >= -(bid-1)},   W_hi = min{v : S(v) >= bid}          (S = signed cents; cents = |S|)
// and, since v = min(mulhi(d, m), 2^24 - 1) is non-decreasing in the word's offset d inside its click / no-click
// sub-interval, it is d in [D(W_lo), D(W_hi)) with D(W) = min{d : v(d) >= W} = ceil(W 2^32 / m).  Per keyword that is two
// word intervals - clicked wins and unclicked wins - and an auction costs two subtract-and-compare pairs instead of a
// logarithm; the competitor's bid itself is only evaluated for the clicked wins, which pay it.  These functions find the
// intervals EXACTLY (the oracle resolves every auction the long way; parity is bit for bit).

// min{v in [0, 2^24] : signed_cents_from_v(v) >= target}.  A float estimate of the answer, a window around it that is
// verified at both ends, bisection inside; if the estimate was off (degenerate parameters) the bisection runs over the
// whole range instead - the result never depends on the estimate.
ADC_HD uint32_t lower_bound_v(int32_t target, float loc, float scale, const LogTableEntry *tab)
{
    const float s = __builtin_fabsf(scale);
    const float x_t = ((float)target - 0.5f) * 0.01f;                    // 100 x rounds to >= target from about here
    const float z_t = (x_t - loc) / s;
    const float e_t = __builtin_fabsf(z_t);
#if defined(__HIP_DEVICE_COMPILE__)
    float mag_f = __builtin_amdgcn_exp2f(fma32(e_t, -1.44269504f, 23.0f));   // u 2^23, u = exp(-e)
#else
    float mag_f = exp2f(fma32(e_t, -1.44269504f, 23.0f));
#endif
    mag_f = mag_f < 8388607.0f ? mag_f : 8388607.0f;                     // (NaN -> the bound; any estimate will do)
    mag_f = mag_f > 0.0f ? mag_f : 0.0f;
    const uint32_t mag = (uint32_t)mag_f;
    const uint32_t est = z_t < 0.0f ? mag : 0x00FFFFFFu - mag;
    // how far the estimate can be off: the table log differs from log by < 4e-6 (x mag), exp2 and its argument by ~1e-6,
    // and one ulp of x = loc + s z moves z by ulp / s
    float slack = mag_f * (6.0e-6f + 2.4e-7f * (__builtin_fabsf(x_t) + __builtin_fabsf(loc)) / s);
    slack = slack < 4194304.0f ? slack : 4194304.0f;                     // (NaN -> the bound)
    const uint32_t half = 8u + (uint32_t)slack;
    uint32_t lo = est > half ? est - half : 0u;
    uint32_t hi = est + half < 0x01000000u ? est + half : 0x01000000u;
    // wanted: S(lo - 1) < target (or lo == 0) and S(hi) >= target (or hi == 2^24)
    if (lo > 0u && !(signed_cents_from_v(lo - 1u, loc, scale, tab) < target)) lo = 0u;
    if (hi < 0x01000000u && !(signed_cents_from_v(hi, loc, scale, tab) >= target)) hi = 0x01000000u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (signed_cents_from_v(mid, loc, scale, tab) >= target) hi = mid;
        else lo = mid + 1u;
    }
    return lo;
}

// min{d in [0, range] : min(mulhi(d, m), 2^24 - 1) >= W} for W in [0, 2^24]; `range` (<= 2^32) if no offset reaches W.
ADC_HD uint64_t offset_reaching(uint32_t W, uint32_t m, uint64_t range)
{
    if (W == 0u) return 0ull;
    if (W > 0x00FFFFFFu || m == 0u) return range;
    const double q = ((double)W * 4294967296.0) / (double)m;             // W 2^32 / m, correctly rounded: within 2^-21 of it
    if (!(q < (double)range)) return range;
    uint64_t d = (uint64_t)q;                                            // floor or ceiling of the quotient
    const uint64_t need = (uint64_t)W << 32;
    if (d * m >= need) { if (d > 0ull && (d - 1ull) * m >= need) d -= 1ull; }
    else { d += 1ull; if (d * m < need) d += 1ull; }
    return d < range ? d : range;
}

// the two word intervals of a keyword: an auction word w is a clicked win iff (w - c_lo) < c_w, an unclicked win iff
// (w - n_lo) < n_w (unsigned 32-bit arithmetic; widths fit because the word 2^32 - 1 never wins)
struct WinIntervals { uint32_t c_lo, c_w, n_lo, n_w; };

ADC_HD WinIntervals win_intervals(int32_t bid_c, float loc, float scale, uint64_t t_click, const AuctionLaw &law, const LogTableEntry *tab)
{
    const uint32_t w_lo = lower_bound_v(1 - bid_c, loc, scale, tab);     // bid > |cents|  <=>  -(bid - 1) <= S <= bid - 1
    const uint32_t w_hi = lower_bound_v(bid_c, loc, scale, tab);
    WinIntervals r{0u, 0u, 0u, 0u};
    if (!(w_lo < w_hi)) return r;
    const uint64_t top = 0xFFFFFFFFull;                                  // words at or above it never win
    uint64_t a = offset_reaching(w_lo, law.m_click, t_click), b = offset_reaching(w_hi, law.m_click, t_click);
    b = b < top ? b : top;
    if (a < b) { r.c_lo = (uint32_t)a; r.c_w = (uint32_t)(b - a); }
    const uint64_t range_n = 4294967296ull - t_click;
    a = t_click + offset_reaching(w_lo, law.m_noclick, range_n);
    b = t_click + offset_reaching(w_hi, law.m_noclick, range_n);
    b = b < top ? b : top;
    if (a < b) { r.n_lo = (uint32_t)a; r.n_w = (uint32_t)(b - a); }
    return r;
}

// ---- conservative brackets of the two win intervals (keyword sets with few auctions per keyword) -------------------
// win_intervals costs several hundred instructions per keyword (two verified bisections, four f64 quotients): right for a
// keyword with a hundred auctions a day, not for one with ten.  win_brackets gives, from the float estimate ALONE (one
// reciprocal, two exp2, no table look-up, no verification), two pairs of word intervals with
//     in.c  subset of  {clicked wins}  subset of  out.c,        in.n  subset of  {unclicked wins}  subset of  out.n
// so a word inside `in` wins, a word outside `out` loses, and only the words in between - a few 1e-5 of all words - have to
// be resolved the long way (auction_outcome_unless_top_word + auction_wins, what the oracle does for every auction).
// Results therefore never depend on the estimate, only the share of words that take the long way does.
//
// Error budget of the estimate v^ of W = lower_bound_v(target), relative to mag = 2^23 u (the table's side of the
// distribution): table log and its float evaluation < 8e-6; the float evaluation of z_t = ((target - 1/2)/100 - loc)/s and
// of x = loc + s z on the law's side 3.6e-7 (|x_t| + |loc|)/s + 1.2e-7 |z_t|; exp2 and its argument < 1.3e-6; + a few
// units for the integer steps.  `slack` below is more than twice that; tests/test_oracle_scalar.py checks the inclusions
// against win_intervals on 10^7 random and adversarial keywords, tests/test_gpu_parity.py on the device's own exp2 / rcp.
// Word space: D(W) = ceil(W 2^32 / m) with m = floor(2^56 / range) good to 2^-23, saturated for range < 2^24, i.e.
// D(W) = W max(range 2^-24, 1) (1 +- 2^-22) + [0, 1); float32 carries words to +-256.  kWordSlack covers both.
struct WinBrackets { WinIntervals out, in; };
constexpr float kWordSlack = 4096.0f;

ADC_HD float fast_exp2(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x);
#else
    return exp2f(x);
#endif
}
ADC_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
ADC_HD float min_num(float a, float b) { return __builtin_fminf(a, b); }       // the non-NaN operand if one is NaN
ADC_HD float max_num(float a, float b) { return __builtin_fmaxf(a, b); }
ADC_HD uint32_t sat_sub(uint32_t a, uint32_t b)                                                      // v_sub_u32 ... clamp
{
#if defined(__clang__)
    return __builtin_elementwise_sub_sat(a, b);
#else
    return a > b ? a - b : 0u;          // (a host build by gcc: oracle/build.py build_shims_host)
#endif
}
// float -> uint32: truncation, saturating at both ends, NaN -> 0 (v_cvt_u32_f32 does exactly this; spelled out for the host)
ADC_HD uint32_t to_word(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)x;
#else
    return !(x > 0.0f) ? 0u : x >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)x;
#endif
}

// v^ and its slack for one target; x_t = (target - 1/2) / 100 is passed in
ADC_HD void lower_bound_estimate(float x_t, float loc, float inv_s, float &est, float &slack)
{
    const float z_t = (x_t - loc) * inv_s;
    const float e_t = __builtin_fabsf(z_t);
    const float mag = min_num(fast_exp2(fma32(e_t, -1.44269504f, 23.0f)), 8388607.0f);
    est = z_t < 0.0f ? mag : 16777215.0f - mag;
    const float rel = fma32(1.0e-6f, fma32(__builtin_fabsf(x_t) + __builtin_fabsf(loc), inv_s, e_t), 2.0e-5f);
    slack = min_num(fma32(mag, rel, 16.0f), 33554432.0f);                   // (NaN -> 2^25: everything is "in between")
}

// t_f = the click threshold T as the float32 it is (bernoulli_threshold_f32(bctr), an integer <= 2^32)
ADC_HD WinBrackets win_brackets(int32_t bid_c, float loc, float scale, float t_f)
{
    const float inv_s = fast_rcp(__builtin_fabsf(scale));
    const float bc = (float)bid_c;
    float lo, lo_s, hi, hi_s;
    lower_bound_estimate(fma32(bc, -0.01f, 0.005f), loc, inv_s, lo, lo_s);      // target 1 - bid_c
    lower_bound_estimate(fma32(bc, 0.01f, -0.005f), loc, inv_s, hi, hi_s);      // target bid_c
    const float lo_out = lo - lo_s, lo_in = lo + lo_s, hi_out = hi + hi_s, hi_in = hi - hi_s;
    const float a_c = max_num(t_f * 5.9604644775390625e-08f, 1.0f);
    const float a_n = max_num((4294967296.0f - t_f) * 5.9604644775390625e-08f, 1.0f);
    const float t_dn = t_f - kWordSlack, t_up = t_f + kWordSlack;
    WinBrackets r;
    // clicked wins live in [0, T): both upper ends stop there
    const uint32_t c_lo_out = to_word(fma32(lo_out, a_c, -kWordSlack)), c_hi_out = to_word(min_num(fma32(hi_out, a_c, kWordSlack), t_f));
    const uint32_t c_lo_in = to_word(fma32(lo_in, a_c, kWordSlack)), c_hi_in = to_word(min_num(fma32(hi_in, a_c, -kWordSlack), t_f));
    // unclicked wins in [T, 2^32 - 1): the conversion saturates at 2^32 - 1, the one word that never wins
    const uint32_t n_lo_out = to_word(fma32(lo_out, a_n, t_dn)), n_hi_out = to_word(fma32(hi_out, a_n, t_up));
    const uint32_t n_lo_in = to_word(fma32(lo_in, a_n, t_up)), n_hi_in = to_word(fma32(hi_in, a_n, t_dn));
    r.out = WinIntervals{c_lo_out, sat_sub(c_hi_out, c_lo_out), n_lo_out, sat_sub(n_hi_out, n_lo_out)};
    r.in = WinIntervals{c_lo_in, sat_sub(c_hi_in, c_lo_in), n_lo_in, sat_sub(n_hi_in, n_lo_in)};
    return r;
}

// the clicked-win interval alone, for a bid: win_intervals(bid_c, ...).c_lo / .c_w.  With bid_c = min(bid, R + 1) these are the
// words whose clicked win costs at most R cents (a win pays the competitor's cents, and wins iff they are below the bid)
ADC_HD void clicked_win_interval(int32_t bid_c, float loc, float scale, uint64_t t_click, const AuctionLaw &law, const LogTableEntry *tab,
                                 uint32_t &lo, uint32_t &width)
{
    lo = 0u;
    width = 0u;
    const uint32_t w_lo = lower_bound_v(1 - bid_c, loc, scale, tab), w_hi = lower_bound_v(bid_c, loc, scale, tab);
    if (!(w_lo < w_hi)) return;
    const uint64_t a = offset_reaching(w_lo, law.m_click, t_click);
    uint64_t b = offset_reaching(w_hi, law.m_click, t_click);
    b = b < 0xFFFFFFFFull ? b : 0xFFFFFFFFull;
    if (a < b) { lo = (uint32_t)a; width = (uint32_t)(b - a); }
}

// revenue of a conversion in cents, round2(max(N(mu, sd), 0.01)) (adcraft/synthetic_kw_helpers.py:66-70), IMPLICIT path
ADC_HD int32_t revenue_cents_tab(uint32_t w, float mu, float sd, const NormTableEntry *tab)
{
    float x = fma32(sd, normal_tab(w, tab), mu);
    x = x > 0.01f ? x : 0.01f;
    return money_to_cents(x);
}

ADC_HD int32_t revenue_cents(uint32_t w, float mu, float sd)
{
    float x = fma32(sd, normal_from_word(w), mu);
    x = x > 0.01f ? x : 0.01f;
    return money_to_cents(x);
}

ADC_HD int32_t volume_from_word(uint32_t w, float mean, float sd, const NormTableEntry *tab)
{
    float x = fma32(sd, normal_tab(w, tab), mean);
    x = x > 0.0f ? x : 0.0f;
    x = x < (float)kVolumeMax ? x : (float)kVolumeMax;
    const float t = __builtin_truncf(x);
    return (int32_t)t + ((x - t) >= 0.5f ? 1 : 0);          // f64::round: half away from zero
}

// action canonicalisation, adcraft/gymnasium_kw_env.py:199,215 (numpy round(x,2) == rint(x*100)/100 in f64)
ADC_HD int64_t bid_to_cents(float bid)
{
    double c = __builtin_rint((double)bid * 100.0);
    if (!(c >= 1.0)) c = 1.0;
    if (c > 1.0e9) c = 1.0e9;
    return (int64_t)c;
}
// the same value as a 32-bit integer (it is at most 1e9): one f64 -> i32 conversion instead of the f64 -> i64 sequence
ADC_HD int32_t bid_to_cents_i32(float bid)
{
    double c = __builtin_rint((double)bid * 100.0);
    if (!(c >= 1.0)) c = 1.0;
    if (c > 1.0e9) c = 1.0e9;
    return (int32_t)c;
}
ADC_HD int64_t budget_to_cents(float budget)
{
    double c = __builtin_rint((double)budget * 100.0);
    if (!(c > -9.0e15)) c = -9.0e15;
    if (c > 9.0e15) c = 9.0e15;
    return (int64_t)c;
}

// (double)cents / 100.0, correctly rounded, without the f64 division sequence: q0 = c * RN(1/100), one exact residual by
// FMA, one correction (Markstein).  Equal to the IEEE quotient for every 0 <= cents < 2^31 - checked exhaustively on the
// CPU by tests/test_oracle_scalar.py::test_cents_to_dollars_is_the_ieee_quotient.
ADC_HD double cents_to_dollars_f64(int cents)
{
    const double a = (double)cents;
    const double q0 = a * 0.01;
    const double rem = __builtin_fma(-q0, 100.0, a);
    return __builtin_fma(rem, 0.01, q0);
}

// (float)cents / 100.0f - the float32 dollars every observation is handed over in - without the division sequence
// while |cents| < 2^24 (the same Markstein step as above, equal to the IEEE quotient for every such value: checked
// exhaustively by tests/test_oracle_scalar.py::test_cents_to_dollars_is_the_ieee_quotient); larger amounts divide.
ADC_HD float cents_to_dollars_f32(long long cents)
{
    const unsigned long long mag = (unsigned long long)(cents < 0 ? -cents : cents);
    if (mag < 16777216ull) {                              // one 64-bit compare, then 32-bit arithmetic
        const float a = (float)(int)cents;
        const float q0 = a * 0.01f;
        return fma32(fma32(-q0, 100.0f, a), 0.01f, q0);
    }
    return (float)cents / 100.0f;
}

// the same for an amount known to be non-negative (a keyword-day's spend or revenue): no sign handling
ADC_HD float cents_to_dollars_f32_nonneg(unsigned long long cents)
{
    if (cents < 16777216ull) {
        unsigned int lo = (unsigned int)cents;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(lo));       // (opaque: knowing the high half to be zero here, hipcc otherwise converts the 64-bit value - 8 instructions for 1)
#endif
        const float a = (float)lo;
        const float q0 = a * 0.01f;
        return fma32(fma32(-q0, 100.0f, a), 0.01f, q0);
    }
    return (float)cents / 100.0f;
}

ADC_HD float clamp01(float v) { v = v > 0.0f ? v : 0.0f; return v < 1.0f ? v : 1.0f; }

ADC_HD float threshold_sigmoid_f32(float bid, float thresh, float intercept, float slope)
{
    const float th = clamp01(2.0f * thresh) / 2.0f;         // halver = 2 + 1e-10 rounds to 2.0f
    const float r = 1.0f / (1.0f + det_exp(-slope * (bid - intercept)));
    return clamp01(fma32(fma32(2.0f, th, 1.0f), r, -th));
}

ADC_HD float explicit_cost(uint32_t w, float bid)
{
    const float sq = __builtin_sqrtf(bid);
    const float sd = 1e-10f + sq / 6.0f;
    float v = fma32(sd, normal_from_word(w), sq / 4.0f + 2.2f);
    v = v > 0.0f ? v : 0.0f;
    return v < 4.4f ? v : 4.4f;
}

// drift coefficient: uniform(-a, a), adcraft/gymnasium_kw_env.py:132-135
ADC_HD float drift_coeff(uint32_t w, float a) { return fma32(2.0f * a, unit_closed24(w), -a); }

// synthetic action: round2(U(lo, hi))
ADC_HD float synthetic_bid(uint32_t w, float lo, float hi)
{
    const float b = fma32(hi - lo, unit_closed24(w), lo);
    return __builtin_rintf(b * 100.0f) / 100.0f;
}

// ---- IMPLICIT_GENERAL, stream revision 4: the number of bidders of a call by INVERSION ------------------------------------
// The reference draws, per keyword and sub-timestep, which of the max_bidders competitors take part (rng.random(n) <= rate,
// synthetic_kw_classes.py:610-621) and only ever uses how many do: B ~ Binomial(max_bidders, rate).  Instead of max_bidders coins
// (8 Philox calls at the default 30) B is read off ONE uniform by walking the pmf up from 0: pmf(0) = q^n, pmf(b + 1) =
// pmf(b) (n - b) / (b + 1) p / q, in float32 built from the deterministic log / exp (so the oracle reproduces every bit; the
// 23-bit uniform resolves probabilities to 1.2e-7).  Applicable while q^n is a normal float and 0 < p < 1; otherwise the coins.
struct BidderLaw {
    float pmf0, ratio;      // q^n and p / q; pmf0 = 0: not applicable (use the coins)
    int n;
};
constexpr int kBidderTable = 128;     // pools up to 127 bidders go by the table (bidders_from_table); the default is 30
ADC_HD BidderLaw general_bidder_law(int max_bidders, float rate)
{
    BidderLaw L{0.0f, 0.0f, max_bidders};
    if (!(rate > 0.0f && rate < 1.0f) || max_bidders < 1 || max_bidders >= kBidderTable) return L;
    const float q = 1.0f - rate;
    const float lp = (float)max_bidders * det_log(q);
    if (!(lp > -60.0f)) return L;
    L.pmf0 = det_exp(lp);
    L.ratio = rate / q;
    return L;
}
ADC_HD int bidders_from_word(uint32_t w, const BidderLaw &L)
{
    const float u = unit_open23(w);
    float pmf = L.pmf0, cdf = L.pmf0;
    int b = 0;
    while (u > cdf && b < L.n) {
        pmf = pmf * ((float)(L.n - b) / (float)(b + 1)) * L.ratio;
        cdf = cdf + pmf;
        ++b;
    }
    return b;
}

// The same walk, done once: cdf[b] = P(B <= b) for b = 0 .. n - 1 exactly as bidders_from_word accumulates it, and the count
// read off by bisection (the sums are non-decreasing) - what the keyword-parallel kernels use: the walk's float division per
// step, at the wave's longest trip count, cost as much as the eight Philox calls it replaced.
ADC_HD void fill_bidder_cdf(float *cdf, const BidderLaw &L)
{
    float pmf = L.pmf0, c = L.pmf0;
    for (int b = 0; b < L.n; ++b) {
        cdf[b] = c;
        pmf = pmf * ((float)(L.n - b) / (float)(b + 1)) * L.ratio;
        c = c + pmf;
    }
}
ADC_HD int bidders_from_table(uint32_t w, const float *cdf, int n)
{
    const float u = unit_open23(w);
    int lo = 0, hi = n;                 // the first b in [0, n) with u <= cdf[b]; n if there is none
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (u > cdf[mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- IMPLICIT_GENERAL (the reference's default ImplicitKeyword), stream revision 3: the top bids as ORDER STATISTICS --------
// An auction of B bidders is cleared against its top (num_winners + 2) competitor bids only (synthetic_kw_helpers.py:152-177).
// Instead of drawing all B Laplace(loc, |scale|) bids (rng.laplace, synthetic_kw_classes.py:681-686: ~18 draws per auction at
// the default pool) and keeping the top few, the top k = min(B, top) are drawn directly, in descending order, from ONE call:
// with E_i ~ Exp(1) independent, t_1 = E_1 / B, t_2 = t_1 + E_2 / (B - 1), ... are -ln of the top uniform order statistics
// (Renyi), and the Laplace quantile at p = exp(-t) is loc + s (ln 2 - t) for p <= 1/2, loc - s ln(2 (1 - p)) above.  Same
// joint law of the top k as sorting B draws (tests/test_oracle_scalar.py checks it against numpy); float32 built from the
// deterministic log / exp above, so the oracle reproduces every bit.  Returns k; out[0] >= out[1] >= ...
ADC_HD int top_laplace_bids(const U4 &w, int B, int top, float loc, float scale, float out[4])
{
    const int k = B < top ? B : top;
    const float s = __builtin_fabsf(scale);
    float t = 0.0f;
    for (int i = 0; i < k && i < 4; ++i) {
        const uint32_t word = i == 0 ? w.x : i == 1 ? w.y : i == 2 ? w.z : w.w;
        const float e = -det_log(unit_open23(word));                 // Exp(1)
        t = t + e / (float)(B - i);
        float z;                                                     // the standard Laplace quantile at exp(-t)
        if (t >= 0.693147182464599609375f) z = 0.693147182464599609375f - t;
        else {
            float q = 1.0f - det_exp(-t);
            q = q > 5.9604644775390625e-08f ? q : 5.9604644775390625e-08f;
            z = -det_log(q + q);
        }
        out[i] = fma32(s, z, loc);
    }
    return k;
}

// nth_price_auction(bid, other_bids, n = 2, num_winners = w) (synthetic_kw_helpers.py:116-180) decided from the top w competitor
// bids alone.  The reference sorts the top (w + 2) bids - zero bids standing in for missing bidders (:156-161) - and wins iff at
// least 3 of them are below the bid, i.e. iff the w-th highest is; the price sorted[index - 1] is the highest of them if even
// that one is below the bid, the w-th highest otherwise.  Nothing below the w-th highest matters, so only min(B, w) order
// statistics are drawn (a prefix of what the oracle draws for its literal sort: same values).  w in {1, 2}.
ADC_HD bool clear_general_auction(const U4 &w4, int B, int num_winners, float loc, float scale, double bid_d, float &price)
{
    float xs[4];
    const int nb = top_laplace_bids(w4, B, num_winners, loc, scale, xs);
    float h1 = -3.0e38f, h2 = -3.0e38f;                  // highest, second highest of bids and stand-in zeros
    for (int h = 0; h < nb; ++h) { const float x = xs[h]; if (x > h1) { h2 = h1; h1 = x; } else if (x > h2) h2 = x; }
    int zeros = num_winners + 2 - B;
    zeros = zeros < 2 ? zeros : 2;
    for (int z = 0; z < zeros; ++z) { if (0.0f > h1) { h2 = h1; h1 = 0.0f; } else if (0.0f > h2) h2 = 0.0f; }
    const float need = num_winners == 1 ? h1 : h2;
    if (!((double)need < bid_d)) return false;           // :167-170 (searchsorted left: a tie loses)
    price = (double)h1 < bid_d ? h1 : need;              // :173-175
    return true;
}

// ---- keyword-set generation on the device (law of sample_implicit_keywords_from_quantile_dfs,
// adcraft/gymnasium_kw_utils.py:295-339 + pull_quantiles_data/quantiles_to_keywords.py:13-28) ------------------
// one quantile table per quantity: B buckets of (min, median, max); value = interp(q, [0, .5, 1], bucket)
struct QuantileTable { const float *mins, *meds, *maxs; int buckets; };

ADC_HD float quantile_sample(const QuantileTable &t, uint32_t w_bucket, uint32_t w_q)
{
    const uint32_t b = mulhi32(w_bucket, (uint32_t)t.buckets);          // rng.integers(0, B)
    const float q = unit_closed24(w_q);                                  // rng.random()
    const float lo = t.mins[b], md = t.meds[b], hi = t.maxs[b];
    return q < 0.5f ? fma32((md - lo) / 0.5f, q, lo) : fma32((hi - md) / 0.5f, q - 0.5f, md);     // np.interp
}

// out[8] in adc_param order.  serial distinguishes successive generations for the same env key.
ADC_HD void generate_implicit_keyword(uint64_t key, uint32_t kw, uint32_t serial, const QuantileTable tab[7], float no_vol_prob,
                                      float out[8])
{
    const uint32_t c3 = 0xFFFF0000u | (serial & 0xFFFFu);
    const U4 a = philox4x32(0u, ST_KEYGEN, kw, c3, (uint32_t)key, (uint32_t)(key >> 32));
    const U4 b = philox4x32(1u, ST_KEYGEN, kw, c3, (uint32_t)key, (uint32_t)(key >> 32));
    const U4 c = philox4x32(2u, ST_KEYGEN, kw, c3, (uint32_t)key, (uint32_t)(key >> 32));
    const U4 d = philox4x32(3u, ST_KEYGEN, kw, c3, (uint32_t)key, (uint32_t)(key >> 32));
    const float v = quantile_sample(tab[0], a.x, a.y);
    const float r = unit_closed24(a.w);
    const bool has = unit_closed24(a.z) > no_vol_prob && v == v;          // :298-300 (NaN volume -> no volume)
    out[0] = has ? __builtin_truncf(v) : 0.0f;                           // (int(v), int(1 + r * 0.5 * v))
    out[1] = has ? __builtin_truncf(fma32(r * 0.5f, v, 1.0f)) : r * 0.5f;
    const float cpc = quantile_sample(tab[1], b.x, b.y);
    const float cpc_sd = quantile_sample(tab[2], b.z, b.w) * cpc;        // std_* are multipliers of the mean (:333-339)
    out[2] = cpc;
    out[3] = cpc_sd > 0.01f ? cpc_sd : 0.01f;
    out[4] = quantile_sample(tab[3], c.x, c.y);
    out[5] = quantile_sample(tab[4], c.z, c.w);
    const float rp = quantile_sample(tab[5], d.x, d.y);
    const float rp_sd = quantile_sample(tab[6], d.z, d.w) * rp;
    out[6] = rp;
    out[7] = rp_sd > 0.01f ? rp_sd : 0.01f;
}

// ---- keyword-set generation on the device, EXPLICIT model (law of sample_random_keywords, adcraft/gymnasium_kw_utils.py:113-156:
// the default-constructor keyword set) -------------------------------------------------------------------------------------
// Every Beta the reference draws there has small integer parameters - (2,5), (5,2), (5,5) - and Beta(a, b) with integer a, b is the
// a-th smallest of a + b - 1 independent uniforms: exact, no rejection loop, no transcendental.  The uniforms are the top 24 bits
// of Philox words (rng.random()'s grid is finer; both are far below any float32 parameter's resolution), so an order statistic
// is itself a 24-bit integer i and the variate is i 2^-24.
template <int N, int KTH>
ADC_HD uint32_t order_statistic24(const uint32_t *w)          // the KTH smallest (1-based) of w[0..N-1] >> 8
{
    uint32_t pick = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t x = w[i] >> 8;
        int rank = 0;                                         // elements before x in a stable ascending sort
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint32_t y = w[j] >> 8;
            rank += (y < x || (y == x && j < i)) ? 1 : 0;
        }
        if (rank == KTH - 1) pick = x;
    }
    return pick;
}
// vol_mean = int(2^x 15 - 1) for x = i 2^-24 (gymnasium_kw_utils.py:129-131; B-8: always 14..28) without evaluating 2^x:
// the value is 14 + #{m in 15..28 : i >= ceil(log2((m + 1) / 15) 2^24)} - the thresholds below, checked against numpy's own float64
// expression on both sides of every boundary (tests/test_oracle_scalar.py)
constexpr int kExplicitVolSteps = 14;
ADC_HD int32_t explicit_vol_mean_from_i24(uint32_t i)
{
    const uint32_t T[kExplicitVolSteps] = {1562117u, 3029500u, 4412986u, 5721651u, 6963174u, 8144111u, 9270101u, 10346029u, 11376159u,
                                           12364231u, 13313546u, 14227028u, 15107285u, 15956650u};
    int32_t v = 14;
#pragma unroll
    for (int m = 0; m < kExplicitVolSteps; ++m) v += i >= T[m] ? 1 : 0;
    return v;
}
constexpr uint32_t kExplicitKeygenFirstCall = 16u;            // (calls 0..3 of ST_KEYGEN belong to generate_implicit_keyword)
constexpr int kExplicitKeygenCalls = 13;
// out[8] in adc_param order: vol_mean, vol_std, impression intercept, impression slope, bctr, sctr, rev_mean, rev_std.
// Words (call c word h = index 4 c + h): 0..5 vol_mean ~ Beta(2,5) | 6 vol_std's uniform | 7 the intercept's | 8..13 sctr ~ Beta(5,2) |
// 16..21 rev_mean ~ 1.5 Beta(2,5) | 24..29 rev_std / rev_mean ~ Beta(2,5) | 32..37 bctr ~ Beta(2,5) | 40..48 slope ~ 25 Beta(5,5)
ADC_HD void generate_explicit_keyword(uint64_t key, uint32_t kw, uint32_t serial, float out[8])
{
    const uint32_t c3 = 0xFFFF0000u | (serial & 0xFFFFu);
    uint32_t w[4 * kExplicitKeygenCalls];
#pragma unroll
    for (int c = 0; c < kExplicitKeygenCalls; ++c) {
        const U4 q = philox4x32(kExplicitKeygenFirstCall + (uint32_t)c, ST_KEYGEN, kw, c3, (uint32_t)key, (uint32_t)(key >> 32));
        w[4 * c] = q.x; w[4 * c + 1] = q.y; w[4 * c + 2] = q.z; w[4 * c + 3] = q.w;
    }
    const float scale24 = 5.9604644775390625e-08f;
    const int32_t vm = explicit_vol_mean_from_i24(order_statistic24<6, 2>(w));                 // :129-131
    out[0] = (float)vm;
    out[1] = (unit_closed24(w[6]) * 0.5f) * (float)(vm + 1);                                    // :133
    out[5] = (float)order_statistic24<6, 5>(w + 8) * scale24;                                   // :135 sctr
    out[2] = unit_closed24(w[7]) * 1.5f;                                                        // :136 imp_intercept
    const float mu = ((float)order_statistic24<6, 2>(w + 16) * scale24) * 1.5f;                // :137
    out[6] = mu;
    out[7] = ((float)order_statistic24<6, 2>(w + 24) * scale24) * mu;                          // :138
    out[4] = (float)order_statistic24<6, 2>(w + 32) * scale24;                                  // :139 bctr
    out[3] = ((float)order_statistic24<9, 5>(w + 40) * scale24) * 25.0f;                       // :140 imp_slope
}

// the 24-way split of a day's volume, adcraft/bidding_simulation.py:151-167
ADC_HD void cell_range(int32_t V, int t, int32_t &j0, int32_t &n)
{
    const int32_t s = V / kTimesteps;
    const int32_t first = V - (kTimesteps - 1) * s;
    if (t == 0) { j0 = 0; n = first; }
    else { j0 = first + (t - 1) * s; n = s; }
}

}  // namespace adc
