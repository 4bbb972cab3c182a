// adc_engine.hip - MI355X (gfx950) vectorised BiddingSimulation step engine: kernels + C ABI.
//
// The path (reference file:line it replaces):
//   BiddingSimulation.step                      adcraft/gymnasium_kw_env.py:160-269
//     simulate_epoch_of_bidding_on_campaign     adcraft/bidding_simulation.py:170-234
//       uniform_get_auctions_per_timestep       adcraft/bidding_simulation.py:151-167
//       simulate_epoch_of_bidding               adcraft/bidding_simulation.py:44-120
//         ImplicitKeyword.auction -> nth_price_auction   adcraft/synthetic_kw_classes.py:623-646,
//                                                        adcraft/synthetic_kw_helpers.py:116-180
//         ExplicitKeyword.auction                adcraft/synthetic_kw_classes.py:493-538, src/lib.rs:54-105
//     update_keywords                           adcraft/gymnasium_kw_env.py:114-158
//
// Kernels
//   k_step_implicit_fast  one workgroup per (env, 256-keyword tile).  Phase 1: one lane per keyword
//        loads its 8 parameters + bid (coalesced SoA), applies pending drift, draws the day's volume V
//        and publishes the keyword's state in LDS.  Phase 2: the tile's auctions are cut into
//        chunks of CH consecutive auctions; chunks are dealt to lanes round-robin (a lane finds
//        its keyword by binary search in the LDS prefix array), so lanes stay busy however V
//        varies between keywords (V=0 keywords cost nothing).  Every auction is one Philox4x32-10
//        call: competitor bid -> 2nd-price clearing -> click -> conversion -> revenue, in integer
//        cents.  Chunk totals go to the keyword's LDS accumulators (integer atomics: order-free, so
//        results do not depend on the lane mapping).  Phase 3: one lane per keyword writes the five
//        observation values (coalesced) and the tile's cost/profit go to the env totals.
//        This pass ignores the budget; it is exact whenever the day's spend stays below it.
//   k_step_exact           one wavefront per env: if the fast pass spent >= budget (or the model is
//        EXPLICIT, or a tape is replayed) it re-runs the env in the reference's exact order
//        (t-major, keyword-minor, click order, with the per-cell break and the campaign stop),
//        64 auctions of a cell at a time with ballot/popcount ranks and a wave prefix sum for the
//        budget walk.  Then the step tail: reward, cumulative profit, truncation, day, termination.
//
// No CPU path exists in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/adcraft_engine.h"
#include "adc_law.h"

#define ADC_EXPORT extern "C" __attribute__((visibility("default")))

namespace adck {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            return fail(_e == hipErrorOutOfMemory ? ADC_ENOMEM : ADC_EHIP,                                \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                               \
        }                                                                                                 \
    } while (0)

constexpr int kFastBlock = 256;   // lanes per workgroup = keywords per tile
constexpr int kChunk = 16;        // auctions per work item
constexpr int kWave = 64;
constexpr int kProfileRing = 2048;   // steps in flight between profile flushes
constexpr int kProfileMarks = 4;     // events per step: before fast | after fast | after exact/tail | after metric

// -------------------------------------------------------------------------------------------------
// device view of the engine (passed by value to kernels)
struct View {
    int N, K, model, max_days;
    double loss_threshold;
    float drift_vol, drift_ctr, drift_cvr;
    int drift_on;
    float imp_thresh;
    int auto_reset;
    int metrics_on;
    // state
    float *params;            // [8][N][K]
    uint64_t *key;            // [N]
    uint32_t *tick;           // [N]
    int32_t *day;             // [N]
    int64_t *cum_cents;       // [N] IMPLICIT
    double *cum;              // [N] EXPLICIT
    uint8_t *drift_pending;   // [N]
    uint8_t *exact_hint;      // [N] budget bound on the previous step: skip the fast pass, go straight to the exact one
    // per-step scratch
    int64_t *env_cost;        // [N] cents spent by the fast pass
    int64_t *env_profit;      // [N]
    // outputs
    int32_t *imp, *clk, *conv;   // [N][K]
    float *cost, *rev;           // [N][K]
    double *reward, *cum_profit; // [N]
    int32_t *day_out;            // [N]
    uint8_t *term, *trunc;       // [N]
    // metrics
    int64_t *metric_profit;      // [K]
    int64_t *metric_scalars;     // [8] (filled on read)
    int64_t *metric_env;         // [4][N] per-env running sums: profit cents, env steps, episodes, truncations
    int64_t *metric_kw;          // [N][K] running sum of keyword profit, cents (metric mode)
    float *flat_obs;             // [N][5K+2] FlatArrayWrapper layout, written after every step when enabled
};

__device__ __forceinline__ float &param_at(const View &v, int p, int env, int k)
{
    return v.params[((size_t)p * v.N + env) * v.K + k];
}

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

__device__ __forceinline__ long long wave_sum_i64(long long x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ long long wave_scan_i64(long long x)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
}
__device__ __forceinline__ int wave_scan_i32(int x)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
}

// log table of the auction law (adc_law.h neg_log_u24), one copy per device, filled at engine creation
__device__ adc::LogTableEntry g_log_table[adc::kLogTableIntervals];

__global__ void k_build_log_table()
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < adc::kLogTableIntervals) g_log_table[i] = adc::log_table_entry(i);
}

// pending drift of one keyword (adcraft/gymnasium_kw_env.py:132-158); `tick_of_draw` = tick of the
// step whose update_keywords() this is
__device__ __forceinline__ void drift_keyword(const View &v, uint64_t key, uint32_t tick_of_draw, int k,
                                              float &vol_mean, float vol_std, float &bctr, float &sctr)
{
    const adc::U4 w = adc::draw(key, 0u, adc::ST_DRIFT, (uint32_t)k, tick_of_draw);
    const float uv = adc::drift_coeff(w.x, v.drift_vol);
    const float uc = adc::drift_coeff(w.y, v.drift_ctr);
    const float us = adc::drift_coeff(w.z, v.drift_cvr);
    const float nv = adc::fma32(uv, vol_std, vol_mean);            // "init volume" is the vol std (:136-137)
    vol_mean = nv > 0.0f ? nv : 0.0f;
    bctr = adc::clamp01(bctr * (1.0f + uc));
    sctr = adc::clamp01(sctr * (1.0f + us));
}

// -------------------------------------------------------------------------------------------------
// FAST PASS (IMPLICIT, engine stream, budget ignored)
// -------------------------------------------------------------------------------------------------
constexpr int kQueueCap = 128;    // per-wave ring of deferred paid clicks (entries), power of two; drained whenever 64 wait

// 72 B per keyword + the 2 KiB log table + the rings = 22 KiB: 7 workgroups (28 waves) per CU's 160 KiB of LDS
struct FastShared {
    int off[kFastBlock];            // exclusive prefix of chunk counts
    int vol[kFastBlock];
    int bid_c[kFastBlock];
    float loc[kFastBlock], scale[kFastBlock], mu[kFastBlock], sd[kFastBlock];
    unsigned int t_click[kFastBlock], m_click[kFastBlock], m_noclick[kFastBlock];     // AuctionLaw
    unsigned int t_conv[kFastBlock];                                                    // saturated conversion threshold
    unsigned int a_imp[kFastBlock], a_clk[kFastBlock], a_conv[kFastBlock];
    unsigned long long a_cost[kFastBlock], a_rev[kFastBlock];
    adc::LogTableEntry logtab[adc::kLogTableIntervals];                                 // copy of g_log_table (kFastBlock == intervals)
    union {
        unsigned int queue[kFastBlock / kWave][kQueueCap];     // phase 2
        int wave_tot[kFastBlock / kWave];                      // phase 1 (before the rings are used)
        long long red[2][kFastBlock / kWave];                  // phase 3 (after they are drained)
    };
};
static_assert(kFastBlock == adc::kLogTableIntervals, "one table entry per lane is copied in phase 1");
static_assert(sizeof(FastShared) <= 163840 / 7, "FastShared must allow 7 workgroups per CU");

__global__ __launch_bounds__(kFastBlock, 7) void k_step_implicit_fast(View v, const float *__restrict__ bids)
{
    __shared__ FastShared sh;
    const int tiles = (v.K + kFastBlock - 1) / kFastBlock;
    const int env = blockIdx.x / tiles;
    const int tile = blockIdx.x - env * tiles;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int k = tile * kFastBlock + tid;
    const bool valid = k < v.K;
    const uint64_t key = v.key[env];
    const uint32_t tick = v.tick[env];
    const bool drift = v.drift_on && v.drift_pending[env];
    if (v.exact_hint[env]) return;      // scheduling hint only: k_step_exact_rows computes this env (same results)

    // ---- phase 1: one lane per keyword ---------------------------------------------------------
    int V = 0;
    if (valid) {
        float vol_mean = param_at(v, ADC_P_VOL_MEAN, env, k);
        const float vol_std = param_at(v, ADC_P_VOL_STD, env, k);
        float bctr = param_at(v, ADC_P_BCTR, env, k);
        float sctr = param_at(v, ADC_P_SCTR, env, k);
        if (drift) {
            drift_keyword(v, key, tick - 1u, k, vol_mean, vol_std, bctr, sctr);
            param_at(v, ADC_P_VOL_MEAN, env, k) = vol_mean;
            param_at(v, ADC_P_BCTR, env, k) = bctr;
            param_at(v, ADC_P_SCTR, env, k) = sctr;
        }
        sh.loc[tid] = param_at(v, ADC_P_A, env, k);
        sh.scale[tid] = param_at(v, ADC_P_B, env, k);
        sh.mu[tid] = param_at(v, ADC_P_REV_MEAN, env, k);
        sh.sd[tid] = param_at(v, ADC_P_REV_STD, env, k);
        const adc::AuctionLaw law = adc::make_auction_law(bctr);
        sh.t_click[tid] = law.t32;
        sh.m_click[tid] = law.m_click;
        sh.m_noclick[tid] = law.m_noclick;
        sh.t_conv[tid] = adc::saturate_threshold(adc::bernoulli_threshold(sctr));
        sh.bid_c[tid] = (int)adc::bid_to_cents(bids[(size_t)env * v.K + k]);
        const adc::U4 w = adc::draw(key, 0u, adc::ST_VOL, (uint32_t)k, tick);
        V = adc::volume_from_word(w.x, vol_mean, vol_std);
    }
    sh.logtab[tid] = g_log_table[tid];
    sh.vol[tid] = V;
    sh.a_imp[tid] = sh.a_clk[tid] = sh.a_conv[tid] = 0u;
    sh.a_cost[tid] = sh.a_rev[tid] = 0ull;
    const int nch = (V + kChunk - 1) / kChunk;
    const int incl = wave_scan_i32(nch);
    if (lane == 63) sh.wave_tot[wv] = incl;
    __syncthreads();
    int wave_base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < kFastBlock / kWave; ++i) {
        const int t = sh.wave_tot[i];
        if (i < wv) wave_base += t;
        total += t;
    }
    sh.off[tid] = wave_base + incl - nch;
    __syncthreads();

    // ---- phase 2: chunks of kChunk auctions, dealt round-robin ----------------------------------
    // Control flow is wave-uniform (idle lanes carry n = 0) so that the deferred-click ring below can
    // be maintained with ballots.  Stage A (every auction): one Philox call per FOUR auctions (one word
    // each) -> click bit + competitor bid, 2nd-price clearing.  Stage B (paid clicks only, ~1/4 of auctions): the
    // click's (keyword, auction) is pushed to a per-wave LDS ring; whenever 64 are waiting, all 64
    // lanes draw the conversion/revenue call together - so the expensive normal-quantile runs on
    // full wavefronts instead of on the ~20 % of lanes that happen to convert.
    unsigned int *const ring = sh.queue[wv];
    unsigned int qhead = 0, qtail = 0;
    const uint32_t kw_base = (uint32_t)(tile * kFastBlock);

    auto resolve_click = [&](unsigned int pos) {
        const unsigned int ent = ring[pos & (kQueueCap - 1)];
        const unsigned int uu = ent >> 24;
        const adc::U4 w2 = adc::draw(key, ent & 0x00FFFFFFu, adc::ST_CONV, kw_base + uu, tick);
        if (adc::bernoulli32(w2.x, sh.t_conv[uu])) {
            const int rv = adc::revenue_cents_bm(w2.y, w2.z, sh.mu[uu], sh.sd[uu], sh.logtab);
            atomicAdd(&sh.a_conv[uu], 1u);
            atomicAdd(&sh.a_rev[uu], (unsigned long long)rv);
        }
    };

    const int rounds = (total + kFastBlock - 1) / kFastBlock;
    for (int r = 0; r < rounds; ++r) {
        const int item = r * kFastBlock + tid;
        int u = 0, j0 = 0, n = 0;
        if (item < total) {
#pragma unroll
            for (int s = kFastBlock / 2; s > 0; s >>= 1)
                if (sh.off[u + s] <= item) u += s;
            j0 = (item - sh.off[u]) * kChunk;
            n = min(kChunk, sh.vol[u] - j0);
        }
        const int bid_c = sh.bid_c[u];
        const float loc = sh.loc[u], scale = sh.scale[u];
        const adc::AuctionLaw law{sh.t_click[u], sh.m_click[u], sh.m_noclick[u]};
        const uint32_t kw = kw_base + (uint32_t)u;
        const unsigned int tag = (unsigned int)u << 24;
        unsigned int imp = 0, clk = 0;
        unsigned long long cost = 0;
        // one auction: 2nd-price clearing against the sampled competitor bid, click, deferred conversion
        auto auction = [&](uint32_t word, int j, bool active) {
            bool click_bit;
            const int comp = adc::auction_outcome(word, law, loc, scale, sh.logtab, click_bit);
            const bool win = active && bid_c > comp;                       // tie loses (helpers.py:167-170)
            const bool click = win && click_bit;
            imp += (unsigned int)win;
            clk += (unsigned int)click;
            cost += click ? (unsigned long long)comp : 0ull;               // 2nd price = the competitor's bid
            const unsigned long long m = __ballot(click);
            const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            if (click) ring[(qtail + rank) & (kQueueCap - 1)] = tag | (unsigned int)j;
            qtail += __popcll(m);
        };
        auto drain = [&]() {
            while (qtail - qhead >= (unsigned int)kWave) {      // checked after every push: <= 63 + 64 entries wait (ring holds 128)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                resolve_click(qhead + lane);
                qhead += kWave;
            }
        };
        for (int i = 0; i < kChunk; i += 4) {
            if (!__any(i < n)) break;
            const adc::U4 w = adc::draw(key, (uint32_t)(j0 + i) >> 2, adc::ST_AUCTION, kw, tick);
            auction(w.x, j0 + i, i < n);
            drain();
            auction(w.y, j0 + i + 1, i + 1 < n);
            drain();
            auction(w.z, j0 + i + 2, i + 2 < n);
            drain();
            auction(w.w, j0 + i + 3, i + 3 < n);
            drain();
        }
        if (imp) atomicAdd(&sh.a_imp[u], imp);
        if (clk) {
            atomicAdd(&sh.a_clk[u], clk);
            atomicAdd(&sh.a_cost[u], cost);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if ((unsigned int)lane < qtail - qhead) resolve_click(qhead + lane);      // fewer than 64 left
    __syncthreads();

    // ---- phase 3: outputs ---------------------------------------------------------------------
    long long my_cost = 0, my_profit = 0;
    if (valid) {
        const size_t o = (size_t)env * v.K + k;
        my_cost = (long long)sh.a_cost[tid];
        const long long r = (long long)sh.a_rev[tid];
        my_profit = r - my_cost;
        v.imp[o] = (int)sh.a_imp[tid];
        v.clk[o] = (int)sh.a_clk[tid];
        v.conv[o] = (int)sh.a_conv[tid];
        v.cost[o] = (float)my_cost / 100.0f;
        v.rev[o] = (float)r / 100.0f;
        if (v.metrics_on) v.metric_kw[o] += my_profit;      // corrected by k_step_exact_rows if it re-runs this env
    }
    my_cost = wave_sum_i64(my_cost);
    my_profit = wave_sum_i64(my_profit);
    if (lane == 0) { sh.red[0][wv] = my_cost; sh.red[1][wv] = my_profit; }
    __syncthreads();
    if (tid == 0) {
        long long c = 0, p = 0;
#pragma unroll
        for (int i = 0; i < kFastBlock / kWave; ++i) { c += sh.red[0][i]; p += sh.red[1][i]; }
        if (tiles == 1) { v.env_cost[env] = c; v.env_profit[env] = p; }
        else {
            atomicAdd((unsigned long long *)&v.env_cost[env], (unsigned long long)c);
            atomicAdd((unsigned long long *)&v.env_profit[env], (unsigned long long)p);
        }
    }
}

// step tail (gymnasium_kw_env.py:222-244), run by one lane per env
__device__ __forceinline__ void step_tail(const View &v, int env, uint32_t tick, bool implicit, long long profit_c, double reward_d)
{
    double reward, cum;
    if (implicit) {
        const long long cc = v.cum_cents[env] + profit_c;
        v.cum_cents[env] = cc;
        reward = (double)profit_c / 100.0;
        cum = (double)cc / 100.0;
    } else {
        reward = reward_d;
        cum = v.cum[env] + reward;
        v.cum[env] = cum;
    }
    const bool truncated = cum < -v.loss_threshold;                 // :225
    const int day = v.day[env] + 1;                                  // :227
    const bool terminated = day >= v.max_days;                       // :228
    v.reward[env] = reward;
    v.cum_profit[env] = cum;
    v.day_out[env] = day;
    v.term[env] = terminated;
    v.trunc[env] = truncated;
    v.day[env] = day;
    v.tick[env] = tick + 1u;
    if (v.drift_on) v.drift_pending[env] = 1;                        // :246 update_keywords()
    if (v.auto_reset && (terminated || truncated)) {
        v.day[env] = 0; v.cum_cents[env] = 0; v.cum[env] = 0.0;      // :327-328
    }
    v.env_cost[env] = 0;
    v.env_profit[env] = 0;
    if (v.metrics_on) {
        // per-env running sums (no same-address atomics: 4096 waves on one word serialise at ~12 ns each)
        const long long pc = implicit ? profit_c : (long long)__double2ll_rn(reward * 100.0);
        v.metric_env[env] += pc;
        v.metric_env[(size_t)v.N + env] += 1;
        if (terminated || truncated) v.metric_env[2 * (size_t)v.N + env] += 1;
        if (truncated) v.metric_env[3 * (size_t)v.N + env] += 1;
    }
}

// -------------------------------------------------------------------------------------------------
// EXACT PASS + STEP TAIL: one wavefront per env
// -------------------------------------------------------------------------------------------------
struct TapeView {
    const int32_t *volumes, *bid_cents, *x_impressions;
    const double *x_cost;
    const uint8_t *click, *conv;
    const int32_t *rev_cents;
    const int64_t *off_bid, *off_ximp, *off_xcost, *off_click, *off_conv, *off_rev;
    int64_t *end_bid, *end_ximp, *end_xcost, *end_click, *end_conv, *end_rev;
    int64_t len_bid, len_ximp, len_xcost, len_click, len_conv, len_rev;
};

// dynamic LDS layout (K entries each): cost i64|f64, rev i64, profit f64, vol i32, imp i32, clk i32, conv i32

template <int MODEL, bool TAPE>
__global__ __launch_bounds__(kWave) void k_step_exact(View v, const float *__restrict__ bids,
                                                       const float *__restrict__ budget_in, TapeView tp, int fast_ran)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int K = v.K;
    long long *s_cost_i = reinterpret_cast<long long *>(lds_raw);                 // [K]
    double *s_cost_d = reinterpret_cast<double *>(lds_raw);                       // [K] (EXPLICIT view)
    long long *s_rev = s_cost_i + K;                                              // [K]
    double *s_profit = reinterpret_cast<double *>(s_rev + K);                     // [K]
    int *s_vol = reinterpret_cast<int *>(s_profit + K);                           // [K]
    int *s_imp = s_vol + K, *s_clk = s_imp + K, *s_conv = s_clk + K;

    const int env = blockIdx.x;
    const int lane = threadIdx.x;
    const uint64_t key = v.key[env];
    const uint32_t tick = v.tick[env];
    const long long budget_c = adc::budget_to_cents(budget_in[env]);

    bool rerun = true;
    if (MODEL == ADC_MODEL_IMPLICIT && !TAPE && fast_ran) rerun = !(v.env_cost[env] < budget_c);

    double reward_d = 0.0;
    long long profit_c = 0;

    if (rerun) {
        // EXPLICIT has no fast pass: pending drift is applied here
        if (MODEL == ADC_MODEL_EXPLICIT && !TAPE && v.drift_on && v.drift_pending[env]) {
            for (int k = lane; k < K; k += kWave) {
                float vm = param_at(v, ADC_P_VOL_MEAN, env, k), bc = param_at(v, ADC_P_BCTR, env, k),
                      sc = param_at(v, ADC_P_SCTR, env, k);
                drift_keyword(v, key, tick - 1u, k, vm, param_at(v, ADC_P_VOL_STD, env, k), bc, sc);
                param_at(v, ADC_P_VOL_MEAN, env, k) = vm;
                param_at(v, ADC_P_BCTR, env, k) = bc;
                param_at(v, ADC_P_SCTR, env, k) = sc;
            }
        }
        for (int k = lane; k < K; k += kWave) {
            int V;
            if (TAPE) V = tp.volumes[(size_t)env * K + k];
            else {
                const adc::U4 w = adc::draw(key, 0u, adc::ST_VOL, (uint32_t)k, tick);
                V = adc::volume_from_word(w.x, param_at(v, ADC_P_VOL_MEAN, env, k), param_at(v, ADC_P_VOL_STD, env, k));
            }
            s_vol[k] = V;
            s_imp[k] = s_clk[k] = s_conv[k] = 0;
            s_cost_i[k] = 0;   // also zeroes the double view
            s_rev[k] = 0;
            s_profit[k] = 0.0;
        }
        __syncthreads();

        long long remaining_c = budget_c;
        double remaining_d = (double)budget_c / 100.0;          // np.round(budget, 2)
        int64_t cur_bid = 0, cur_ximp = 0, cur_xcost = 0, cur_click = 0, cur_conv = 0, cur_rev = 0;
        if (TAPE) {
            cur_bid = tp.off_bid ? tp.off_bid[env] : 0;
            cur_ximp = tp.off_ximp ? tp.off_ximp[env] : 0;
            cur_xcost = tp.off_xcost ? tp.off_xcost[env] : 0;
            cur_click = tp.off_click ? tp.off_click[env] : 0;
            cur_conv = tp.off_conv ? tp.off_conv[env] : 0;
            cur_rev = tp.off_rev ? tp.off_rev[env] : 0;
        }
        bool stop = false;
        const unsigned long long lt = lanemask_lt();

        for (int t = 0; t < adc::kTimesteps && !stop; ++t) {
            for (int k = 0; k < K && !stop; ++k) {
                int32_t j0, n;
                adc::cell_range(s_vol[k], t, j0, n);
                const long long bid_c = adc::bid_to_cents(bids[(size_t)env * K + k]);
                const uint64_t t_click = adc::bernoulli_threshold(param_at(v, ADC_P_BCTR, env, k));
                const uint64_t t_conv = adc::bernoulli_threshold(param_at(v, ADC_P_SCTR, env, k));
                const float mu = param_at(v, ADC_P_REV_MEAN, env, k), sd = param_at(v, ADC_P_REV_STD, env, k);

                if (MODEL == ADC_MODEL_IMPLICIT) {
                    const float loc = param_at(v, ADC_P_A, env, k), scale = param_at(v, ADC_P_B, env, k);
                    const adc::AuctionLaw law = adc::make_auction_law(param_at(v, ADC_P_BCTR, env, k));
                    int wins = 0, paid = 0, convs = 0;
                    long long clicked_sum = 0, cell_cost = 0, cell_rev = 0;
                    for (int base = 0; base < n; base += kWave) {
                        const int i = base + lane;
                        const bool act = i < n;
                        bool click_bit = false;
                        long long comp = 0;
                        const uint32_t j = (uint32_t)(j0 + i);
                        if (act) {
                            if (TAPE) comp = tp.bid_cents[cur_bid + i];
                            else {
                                const adc::U4 w = adc::draw(key, j >> 2, adc::ST_AUCTION, (uint32_t)k, tick);
                                const uint32_t word = (j & 3u) == 0 ? w.x : (j & 3u) == 1 ? w.y : (j & 3u) == 2 ? w.z : w.w;
                                comp = adc::auction_outcome(word, law, loc, scale, g_log_table, click_bit);
                            }
                        }
                        const bool win = act && bid_c > comp;
                        const unsigned long long win_mask = __ballot(win);
                        bool clicked = false;
                        if (win) {
                            if (TAPE) clicked = tp.click[cur_click + wins + __popcll(win_mask & lt)] != 0;
                            else clicked = click_bit;
                        }
                        // budget walk (bidding_simulation.py:97-104): click i is paid iff the running
                        // sum of clicked costs up to and including it fits the cell's opening budget
                        const long long x = clicked ? comp : 0;
                        const long long pre = clicked_sum + wave_scan_i64(x);
                        const bool is_paid = clicked && pre <= remaining_c;
                        const unsigned long long paid_mask = __ballot(is_paid);
                        bool convd = false;
                        long long rv = 0;
                        if (is_paid) {
                            if (TAPE) convd = tp.conv[cur_conv + paid + __popcll(paid_mask & lt)] != 0;
                            else {
                                const adc::U4 w2 = adc::draw(key, j, adc::ST_CONV, (uint32_t)k, tick);
                                convd = adc::bernoulli(w2.x, t_conv);
                                if (convd) rv = adc::revenue_cents_bm(w2.y, w2.z, mu, sd, g_log_table);
                            }
                        }
                        const unsigned long long conv_mask = __ballot(convd);
                        if (TAPE && convd) rv = tp.rev_cents[cur_rev + convs + __popcll(conv_mask & lt)];
                        clicked_sum += wave_sum_i64(x);
                        cell_cost += wave_sum_i64(is_paid ? comp : 0);
                        cell_rev += wave_sum_i64(rv);
                        wins += __popcll(win_mask);
                        paid += __popcll(paid_mask);
                        convs += __popcll(conv_mask);
                    }
                    if (TAPE) { cur_bid += n; cur_click += wins; cur_conv += paid; cur_rev += convs; }
                    if (lane == 0) {
                        s_imp[k] += wins;          // impressions are not budget-limited (:86-88)
                        s_clk[k] += paid;
                        s_conv[k] += convs;
                        s_cost_i[k] += cell_cost;
                        s_rev[k] += cell_rev;
                    }
                    remaining_c -= cell_cost;      // :225
                    if (remaining_c <= 0) stop = true;   // :230-233
                } else {
                    // EXPLICIT cell
                    const float bid_d = (float)((double)bid_c / 100.0);
                    const float p_imp = adc::threshold_sigmoid_f32(bid_d, v.imp_thresh, param_at(v, ADC_P_A, env, k),
                                                                   param_at(v, ADC_P_B, env, k));
                    const uint64_t t_imp = adc::bernoulli_threshold(p_imp);
                    double budget = remaining_d, cell_cost = 0.0;
                    double obs_cost = s_cost_d[k];           // obs cost = sum_list(costs): left to right
                    long long cell_rev_c = 0;
                    int imps = 0, paid = 0, convs = 0;
                    bool broke = false;
                    int entries = n;
                    if (TAPE) entries = tp.x_impressions[cur_ximp];
                    for (int base = 0; base < entries; base += kWave) {
                        const int i = base + lane;
                        const bool act = i < entries;
                        adc::U4 w{0u, 0u, 0u, 0u};
                        bool is_imp = false, clicked = false;
                        double cost = 0.0;
                        if (act) {
                            if (TAPE) {
                                is_imp = true;
                                cost = tp.x_cost[cur_xcost + i];
                                clicked = tp.click[cur_click + i] != 0;
                            } else {
                                w = adc::draw(key, (uint32_t)(j0 + i), adc::ST_AUCTION, (uint32_t)k, tick);
                                is_imp = adc::bernoulli(w.x, t_imp);          // Binomial(n,p) as n Bernoullis
                                cost = (double)adc::explicit_cost(w.y, bid_d);
                                clicked = is_imp && adc::bernoulli(w.z, t_click);
                            }
                        }
                        imps += __popcll(__ballot(is_imp));
                        // the reference walks the clicks one by one in f64 (:97-104): same order, same ops
                        unsigned long long cand = __ballot(clicked);
                        unsigned long long paid_mask = 0ull;
                        while (cand && !broke) {
                            const int l = __ffsll((long long)cand) - 1;
                            cand &= cand - 1ull;
                            const double c = __shfl(cost, l, 64);
                            if (budget >= c) { budget -= c; cell_cost += c; obs_cost += c; paid_mask |= 1ull << l; }
                            else broke = true;
                        }
                        const bool is_paid = (paid_mask >> lane) & 1ull;
                        bool convd = false;
                        if (is_paid) {
                            if (TAPE) convd = tp.conv[cur_conv + paid + __popcll(paid_mask & lt)] != 0;
                            else convd = adc::bernoulli(w.w, t_conv);
                        }
                        const unsigned long long conv_mask = __ballot(convd);
                        long long rv = 0;
                        if (convd) {
                            if (TAPE) rv = tp.rev_cents[cur_rev + convs + __popcll(conv_mask & lt)];
                            else {
                                const adc::U4 w2 = adc::draw(key, (uint32_t)(j0 + i), adc::ST_XREV, (uint32_t)k, tick);
                                rv = adc::revenue_cents(w2.x, mu, sd);
                            }
                        }
                        cell_rev_c += wave_sum_i64(rv);
                        paid += __popcll(paid_mask);
                        convs += __popcll(conv_mask);
                    }
                    if (TAPE) {
                        cur_ximp += 1;
                        cur_xcost += entries;
                        cur_click += entries;
                        cur_conv += paid;
                        cur_rev += convs;
                    }
                    if (imps == 0) {
                        // phantom zero-cost click opportunity (synthetic_kw_classes.py:514-515), once per cell
                        bool clicked, convd = false;
                        long long rv = 0;
                        adc::U4 w{0u, 0u, 0u, 0u};
                        if (TAPE) clicked = tp.click[cur_click++] != 0;
                        else {
                            w = adc::draw(key, (uint32_t)t, adc::ST_XPHANTOM, (uint32_t)k, tick);
                            clicked = adc::bernoulli(w.x, t_click);
                        }
                        if (clicked && budget >= 0.0) {
                            ++paid;
                            if (TAPE) convd = tp.conv[cur_conv++] != 0;
                            else convd = adc::bernoulli(w.y, t_conv);
                            if (convd) {
                                ++convs;
                                if (TAPE) rv = tp.rev_cents[cur_rev++];
                                else rv = adc::revenue_cents(w.z, mu, sd);
                                cell_rev_c += rv;
                            }
                        }
                    }
                    if (lane == 0) {
                        s_imp[k] += imps;
                        s_clk[k] += paid;
                        s_conv[k] += convs;
                        s_cost_d[k] = obs_cost;
                        s_rev[k] += cell_rev_c;
                        s_profit[k] += (double)cell_rev_c / 100.0 - cell_cost;    // combine_outcomes: profit += profit
                    }
                    remaining_d -= cell_cost;
                    if (remaining_d <= 0.0) stop = true;
                }
            }
        }
        __syncthreads();
        if (TAPE && lane == 0) {
            if (tp.end_bid) tp.end_bid[env] = cur_bid;
            if (tp.end_ximp) tp.end_ximp[env] = cur_ximp;
            if (tp.end_xcost) tp.end_xcost[env] = cur_xcost;
            if (tp.end_click) tp.end_click[env] = cur_click;
            if (tp.end_conv) tp.end_conv[env] = cur_conv;
            if (tp.end_rev) tp.end_rev[env] = cur_rev;
        }
        // per-keyword observations of the exact pass
        long long pc = 0;
        for (int k = lane; k < K; k += kWave) {
            const size_t o = (size_t)env * K + k;
            v.imp[o] = s_imp[k];
            v.clk[o] = s_clk[k];
            v.conv[o] = s_conv[k];
            const long long r = s_rev[k];
            v.rev[o] = (float)r / 100.0f;
            if (MODEL == ADC_MODEL_IMPLICIT) {
                const long long c = s_cost_i[k];
                v.cost[o] = (float)c / 100.0f;
                pc += r - c;
            } else {
                v.cost[o] = (float)s_cost_d[k];
            }
        }
        profit_c = wave_sum_i64(pc);
        if (MODEL == ADC_MODEL_EXPLICIT) {
            // reward = rust.sum_list(profit_k): left to right (gymnasium_kw_env.py:222)
            double r = 0.0;
            for (int k = 0; k < K; ++k) r += s_profit[k];
            reward_d = r;
        }
    } else {
        profit_c = v.env_profit[env];
    }

    if (lane == 0) step_tail(v, env, tick, MODEL == ADC_MODEL_IMPLICIT, profit_c, reward_d);
}

// -------------------------------------------------------------------------------------------------
// EXACT PASS, ROW-PARALLEL (IMPLICIT, engine stream): one 256-lane workgroup per env
// -------------------------------------------------------------------------------------------------
// The reference walks the day's 24 x K cells in order (t-major, keyword-minor) with one shared budget
// (adcraft/bidding_simulation.py:214-233).  Per sub-timestep row t:
//   pass A  every cell (t,k) of the row is evaluated in parallel (one lane per cell): wins W, clicked
//           wins NC, their total cost T, and the cost X1 of the first clicked win;
//   resolve the budget walk over the row's cells, in keyword order, as a loop of three block-wide steps:
//           (a) bulk - cells whose running total stays strictly below the remaining budget are paid in
//               full (every click affordable, remaining stays > 0);
//           (b) the first cell that does not fit is walked click by click by its lane (pay while the
//               running cost fits, :97-104); exact exhaustion (remaining == 0) stops the campaign (:230-233);
//           (c) skip - with what is left, a cell whose FIRST click is unaffordable pays nothing (the
//               reference breaks at it), so jump to the next cell with X1 <= remaining;
//   pass B  cells paid in full are re-walked with the conversion / revenue draws; the others only count
//           their impressions (impressions are not budget-limited, :86-88).
// Costs are >= 0, so remaining never increases and (a)/(c) are exact, not heuristics.
constexpr int kRowsBlock = 256;
constexpr int kRowsMaxK = 1024;
constexpr int kRowsRing = 128;        // per-wave ring of paid clicks awaiting their conversion / revenue draw

struct CellStat {
    unsigned int wins, clicks, first;     // first = cost of the first clicked win (0xFFFFFFFF if none)
    unsigned int mask;                    // bit i = auction j0+i of the cell is a clicked win (cells of <= 32 auctions)
    unsigned long long total;             // cost of all clicked wins
};

// mode 0: statistics only.  mode 1: every click is paid (adds conversions / revenue).  mode 2: pay while the
// running cost fits `budget` (the reference's in-cell loop); `paid_cost` returns what was spent.
template <int MODE>
__device__ __forceinline__ CellStat walk_cell(uint64_t key, uint32_t tick, uint32_t kw, int j0, int n, int bid_c, float loc,
                                              float scale, const adc::AuctionLaw &law, unsigned long long t_conv, float mu,
                                              float sd, long long budget, unsigned int &conv_out,
                                              unsigned long long &rev_out, unsigned long long &paid_cost)
{
    CellStat st{0u, 0u, 0xFFFFFFFFu, 0u, 0ull};
    conv_out = 0u;
    rev_out = 0ull;
    paid_cost = 0ull;
    bool broke = false;
    const int jend = j0 + n;
    for (int q = j0 >> 2; 4 * q < jend; ++q) {
        const adc::U4 w = adc::draw(key, (uint32_t)q, adc::ST_AUCTION, kw, tick);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int j = 4 * q + h;
            if (j < j0 || j >= jend) continue;
            bool click_bit;
            const int comp = adc::auction_outcome(h == 0 ? w.x : h == 1 ? w.y : h == 2 ? w.z : w.w, law, loc, scale, g_log_table, click_bit);
            if (!(bid_c > comp)) continue;
            st.wins += 1u;
            if (!click_bit) continue;
            if (MODE == 0) {
                st.clicks += 1u;
                st.total += (unsigned long long)comp;
                if (st.first == 0xFFFFFFFFu) st.first = (unsigned int)comp;
                if (j - j0 < 32) st.mask |= 1u << (j - j0);
                continue;
            }
            if (MODE == 2) {
                if (broke) continue;
                if ((long long)(paid_cost + (unsigned long long)comp) > budget) { broke = true; continue; }
            }
            st.clicks += 1u;
            paid_cost += (unsigned long long)comp;
            const adc::U4 w2 = adc::draw(key, (uint32_t)j, adc::ST_CONV, kw, tick);
            if (adc::bernoulli(w2.x, t_conv)) {
                conv_out += 1u;
                rev_out += (unsigned long long)adc::revenue_cents_bm(w2.y, w2.z, mu, sd, g_log_table);
            }
        }
    }
    st.total = MODE == 0 ? st.total : paid_cost;
    return st;
}

struct RowsShared {
    long long wave_part[kRowsBlock / kWave];
    int wave_min[kRowsBlock / kWave];
    long long remaining;
    long long carry;
    int cur;
    int found;
    int stopped;
};

// block-wide: smallest index in [lo, K) whose predicate holds, K if none.  pred is evaluated per lane.
template <typename Pred>
__device__ __forceinline__ int block_first(RowsShared &rs, int lo, int K, Pred pred)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int best = K;
    for (int k = lo + tid; k < K; k += kRowsBlock)
        if (pred(k)) { best = k; break; }        // a lane's indices ascend, the first hit is its smallest
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
    __syncthreads();
    if (lane == 0) rs.wave_min[wv] = best;
    __syncthreads();
    int r = rs.wave_min[0];
#pragma unroll
    for (int i = 1; i < kRowsBlock / kWave; ++i) r = min(r, rs.wave_min[i]);
    return r;
}

__global__ __launch_bounds__(kRowsBlock) void k_step_exact_rows(View v, const float *__restrict__ bids,
                                                                const float *__restrict__ budget_in)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ RowsShared rs;
    const int K = v.K;
    // per keyword (persistent over the day)
    unsigned long long *a_cost = reinterpret_cast<unsigned long long *>(lds_raw);        // [K]
    unsigned long long *a_rev = a_cost + K;                                               // [K]
    unsigned long long *c_total = a_rev + K;          // per cell of the current row: cost of clicked wins
    unsigned long long *c_prefix = c_total + K;       // inclusive running total from `cur`
    int *s_vol = reinterpret_cast<int *>(c_prefix + K);
    unsigned int *a_imp = reinterpret_cast<unsigned int *>(s_vol + K);
    unsigned int *a_clk = a_imp + K, *a_conv = a_clk + K;
    unsigned int *c_wins = a_conv + K, *c_first = c_wins + K, *c_mask = c_first + K, *c_clicks = c_mask + K;
    unsigned int *rings = c_clicks + K;                                         // [kRowsBlock / kWave][kRowsRing]
    unsigned char *c_state = reinterpret_cast<unsigned char *>(rings + (kRowsBlock / kWave) * kRowsRing);
    // c_state: 0 impressions only, 1 paid in full, 2 done, 3 not visited

    const int env = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t key = v.key[env];
    const uint32_t tick = v.tick[env];
    const long long budget_c = adc::budget_to_cents(budget_in[env]);

    const bool hinted = v.exact_hint[env] != 0;
    if (!hinted && v.env_cost[env] < budget_c) {      // the fast pass was exact for this env: only the tail remains
        if (tid == 0) step_tail(v, env, tick, true, v.env_profit[env], 0.0);
        return;
    }
    const bool drift = hinted && v.drift_on && v.drift_pending[env];    // the skipped fast pass would have applied it
    bool was_binding = false;

    for (int k = tid; k < K; k += kRowsBlock) {
        if (drift) {
            float vm = param_at(v, ADC_P_VOL_MEAN, env, k), bc = param_at(v, ADC_P_BCTR, env, k), sc = param_at(v, ADC_P_SCTR, env, k);
            drift_keyword(v, key, tick - 1u, k, vm, param_at(v, ADC_P_VOL_STD, env, k), bc, sc);
            param_at(v, ADC_P_VOL_MEAN, env, k) = vm;
            param_at(v, ADC_P_BCTR, env, k) = bc;
            param_at(v, ADC_P_SCTR, env, k) = sc;
        }
        const adc::U4 w = adc::draw(key, 0u, adc::ST_VOL, (uint32_t)k, tick);
        s_vol[k] = adc::volume_from_word(w.x, param_at(v, ADC_P_VOL_MEAN, env, k), param_at(v, ADC_P_VOL_STD, env, k));
        a_imp[k] = a_clk[k] = a_conv[k] = 0u;
        a_cost[k] = a_rev[k] = 0ull;
    }
    if (tid == 0) { rs.remaining = budget_c; rs.stopped = 0; }
    __syncthreads();

    for (int t = 0; t < adc::kTimesteps; ++t) {
        // ---- pass A: statistics of every cell of the row
        long long my_total = 0;
        for (int k = tid; k < K; k += kRowsBlock) {
            int32_t j0, n;
            adc::cell_range(s_vol[k], t, j0, n);
            unsigned int cv; unsigned long long rv, pc;
            const CellStat st = walk_cell<0>(key, tick, (uint32_t)k, j0, n, (int)adc::bid_to_cents(bids[(size_t)env * K + k]),
                                             param_at(v, ADC_P_A, env, k), param_at(v, ADC_P_B, env, k),
                                             adc::make_auction_law(param_at(v, ADC_P_BCTR, env, k)), 0ull, 0.f, 0.f, 0, cv, rv, pc);
            c_wins[k] = st.wins;
            c_first[k] = st.first;
            c_total[k] = st.total;
            c_mask[k] = st.mask;
            c_clicks[k] = st.clicks;
            c_state[k] = 1;
            my_total += (long long)st.total;
        }
        my_total = wave_sum_i64(my_total);
        if (lane == 0) rs.wave_part[wv] = my_total;
        __syncthreads();
        long long row_total = 0;
#pragma unroll
        for (int i = 0; i < kRowsBlock / kWave; ++i) row_total += rs.wave_part[i];
        long long R = rs.remaining;
        __syncthreads();

        if (!(R - row_total > 0)) {
            // ---- the row does not fit as a whole: resolve it in keyword order
            was_binding = true;
            int cur = 0;
            bool stop = false;
            while (cur < K && !stop) {
                // (a) bulk: running total from `cur`; cells with prefix < R are paid in full
                long long carry = 0;
                int kfull = K;
                for (int base = cur; base < K && kfull == K; base += kRowsBlock) {
                    const int k = base + tid;
                    const long long x = k < K ? (long long)c_total[k] : 0;
                    long long incl = wave_scan_i64(x);
                    if (lane == 63) rs.wave_part[wv] = incl;
                    __syncthreads();
                    long long wbase = carry;
                    for (int i = 0; i < wv; ++i) wbase += rs.wave_part[i];
                    long long tile_total = 0;
                    for (int i = 0; i < kRowsBlock / kWave; ++i) tile_total += rs.wave_part[i];
                    incl += wbase;
                    if (k < K) c_prefix[k] = (unsigned long long)incl;
                    __syncthreads();
                    kfull = block_first(rs, base, min(K, base + kRowsBlock), [&](int kk) { return !((long long)c_prefix[kk] < R); });
                    if (kfull >= min(K, base + kRowsBlock)) kfull = K;
                    carry += tile_total;
                }
                // cells [cur, kfull) keep state 1 (paid in full)
                if (kfull > cur) R -= (long long)c_prefix[kfull - 1];
                __syncthreads();
                cur = kfull;
                if (cur >= K) break;
                // (b) the cell that does not fit: its lane walks it click by click
                if (tid == (cur % kRowsBlock)) {
                    const int k = cur;
                    int32_t j0, n;
                    adc::cell_range(s_vol[k], t, j0, n);
                    unsigned int cv; unsigned long long rv, pc;
                    const CellStat st = walk_cell<2>(key, tick, (uint32_t)k, j0, n, (int)adc::bid_to_cents(bids[(size_t)env * K + k]),
                                                     param_at(v, ADC_P_A, env, k), param_at(v, ADC_P_B, env, k),
                                                     adc::make_auction_law(param_at(v, ADC_P_BCTR, env, k)),
                                                     adc::bernoulli_threshold(param_at(v, ADC_P_SCTR, env, k)),
                                                     param_at(v, ADC_P_REV_MEAN, env, k), param_at(v, ADC_P_REV_STD, env, k), R, cv, rv, pc);
                    a_imp[k] += st.wins;
                    a_clk[k] += st.clicks;
                    a_cost[k] += pc;
                    atomicAdd(&a_conv[k], cv);
                    atomicAdd(&a_rev[k], rv);
                    c_state[k] = 2;
                    rs.carry = (long long)pc;
                }
                __syncthreads();
                R -= rs.carry;
                __syncthreads();
                if (R <= 0) {                      // campaign stop (:230-233): later cells are never visited
                    stop = true;
                    for (int k = cur + 1 + tid; k < K; k += kRowsBlock) c_state[k] = 3;
                    break;
                }
                cur += 1;
                // (c) skip cells whose first click is unaffordable (they pay nothing, impressions still count)
                const int knext = block_first(rs, cur, K, [&](int kk) { return (long long)c_first[kk] <= R; });
                for (int k = cur + tid; k < knext; k += kRowsBlock) c_state[k] = 0;
                __syncthreads();
                cur = knext;
            }
            if (stop && tid == 0) rs.stopped = 1;
        } else {
            R -= row_total;
        }
        __syncthreads();
        if (tid == 0) rs.remaining = R;
        // ---- pass B: commit.  Cells paid in full take their statistics as they are; each of their clicked wins
        // (a bit of the cell's mask) goes through a per-wave ring so that the conversion / revenue call runs on
        // full wavefronts (as in k_step_implicit_fast).  Cells of more than 32 auctions are re-walked instead.
        {
            unsigned int *const ring = rings + wv * kRowsRing;
            unsigned int qhead = 0, qtail = 0;
            auto resolve = [&](unsigned int pos) {
                const unsigned int ent = ring[pos & (kRowsRing - 1)];
                const unsigned int kk = ent >> 20, j = ent & 0x000FFFFFu;          // K <= 1024, j < 2^20
                const adc::U4 w2 = adc::draw(key, j, adc::ST_CONV, kk, tick);
                if (adc::bernoulli(w2.x, adc::bernoulli_threshold(param_at(v, ADC_P_SCTR, env, kk)))) {
                    const int rv = adc::revenue_cents_bm(w2.y, w2.z, param_at(v, ADC_P_REV_MEAN, env, kk),
                                                         param_at(v, ADC_P_REV_STD, env, kk), g_log_table);
                    atomicAdd(&a_conv[kk], 1u);
                    atomicAdd(&a_rev[kk], (unsigned long long)rv);
                }
            };
            for (int kb = 0; kb < K; kb += kRowsBlock) {             // wave-uniform trip count
                const int k = kb + tid;
                unsigned int bits = 0;
                int j0 = 0;
                if (k < K) {
                    const unsigned char stt = c_state[k];
                    int32_t n;
                    adc::cell_range(s_vol[k], t, j0, n);
                    if (stt == 0) a_imp[k] += c_wins[k];
                    else if (stt == 1 && n <= 32) {
                        a_imp[k] += c_wins[k];
                        a_clk[k] += c_clicks[k];
                        a_cost[k] += c_total[k];
                        bits = c_mask[k];
                    } else if (stt == 1) {
                        unsigned int cv; unsigned long long rv, pc;
                        const CellStat st = walk_cell<1>(key, tick, (uint32_t)k, j0, n, (int)adc::bid_to_cents(bids[(size_t)env * K + k]),
                                                         param_at(v, ADC_P_A, env, k), param_at(v, ADC_P_B, env, k),
                                                         adc::make_auction_law(param_at(v, ADC_P_BCTR, env, k)),
                                                         adc::bernoulli_threshold(param_at(v, ADC_P_SCTR, env, k)),
                                                         param_at(v, ADC_P_REV_MEAN, env, k), param_at(v, ADC_P_REV_STD, env, k), 0, cv, rv, pc);
                        a_imp[k] += st.wins;
                        a_clk[k] += st.clicks;
                        a_cost[k] += pc;
                        atomicAdd(&a_conv[k], cv);
                        atomicAdd(&a_rev[k], rv);
                    }
                }
                while (__any(bits != 0u)) {                            // every lane pushes its lowest remaining click
                    const bool has = bits != 0u;
                    const unsigned long long m = __ballot(has);
                    const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                    if (has) {
                        const int i = __ffs((int)bits) - 1;
                        bits &= bits - 1u;
                        ring[(qtail + rank) & (kRowsRing - 1)] = ((unsigned int)k << 20) | (unsigned int)(j0 + i);
                    }
                    qtail += __popcll(m);
                    while (qtail - qhead >= (unsigned int)kWave) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        resolve(qhead + lane);
                        qhead += kWave;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if ((unsigned int)lane < qtail - qhead) resolve(qhead + lane);
        }
        __syncthreads();
        if (rs.stopped) break;
    }

    // ---- observations and the step tail
    long long pc = 0;
    for (int k = tid; k < K; k += kRowsBlock) {
        const size_t o = (size_t)env * K + k;
        const long long c = (long long)a_cost[k], r = (long long)a_rev[k];
        if (v.metrics_on) {
            long long old = 0;          // what the (discarded) fast pass added for this keyword, if it ran
            if (!hinted) old = (long long)__builtin_rintf(v.rev[o] * 100.0f) - (long long)__builtin_rintf(v.cost[o] * 100.0f);
            v.metric_kw[o] += (r - c) - old;
        }
        v.imp[o] = (int)a_imp[k];
        v.clk[o] = (int)a_clk[k];
        v.conv[o] = (int)a_conv[k];
        v.cost[o] = (float)c / 100.0f;
        v.rev[o] = (float)r / 100.0f;
        pc += r - c;
    }
    pc = wave_sum_i64(pc);
    __syncthreads();
    if (lane == 0) rs.wave_part[wv] = pc;
    __syncthreads();
    if (tid == 0) {
        long long profit_c = 0;
        for (int i = 0; i < kRowsBlock / kWave; ++i) profit_c += rs.wave_part[i];
        v.exact_hint[env] = was_binding ? 1 : 0;
        step_tail(v, env, tick, true, profit_c, 0.0);
    }
}

// -------------------------------------------------------------------------------------------------
// small kernels
// -------------------------------------------------------------------------------------------------
__global__ void k_materialize_drift(View v)
{
    const int env = blockIdx.x;
    if (!(v.drift_on && v.drift_pending[env])) return;
    const uint64_t key = v.key[env];
    const uint32_t tick = v.tick[env];
    for (int k = threadIdx.x; k < v.K; k += blockDim.x) {
        float vm = param_at(v, ADC_P_VOL_MEAN, env, k), bc = param_at(v, ADC_P_BCTR, env, k), sc = param_at(v, ADC_P_SCTR, env, k);
        drift_keyword(v, key, tick - 1u, k, vm, param_at(v, ADC_P_VOL_STD, env, k), bc, sc);
        param_at(v, ADC_P_VOL_MEAN, env, k) = vm;
        param_at(v, ADC_P_BCTR, env, k) = bc;
        param_at(v, ADC_P_SCTR, env, k) = sc;
    }
    __syncthreads();
    if (threadIdx.x == 0) v.drift_pending[env] = 0;
}

// update_keywords() called by the user: schedule one more drift now (draw keyed by the current tick)
__global__ void k_force_drift(View v)
{
    const int env = blockIdx.x;
    const uint64_t key = v.key[env];
    const uint32_t tick = v.tick[env];
    for (int k = threadIdx.x; k < v.K; k += blockDim.x) {
        float vm = param_at(v, ADC_P_VOL_MEAN, env, k), bc = param_at(v, ADC_P_BCTR, env, k), sc = param_at(v, ADC_P_SCTR, env, k);
        drift_keyword(v, key, tick, k, vm, param_at(v, ADC_P_VOL_STD, env, k), bc, sc);
        param_at(v, ADC_P_VOL_MEAN, env, k) = vm;
        param_at(v, ADC_P_BCTR, env, k) = bc;
        param_at(v, ADC_P_SCTR, env, k) = sc;
    }
    __syncthreads();
    if (threadIdx.x == 0) v.tick[env] = tick + 1u;    // the draw must not be reused by the next step
}

// metric_scalars[q] = sum over envs of metric_env[q][env], q = blockIdx.x (read-time reduction)
__global__ void k_metric_reduce(View v)
{
    __shared__ long long part[256 / kWave];
    const int q = blockIdx.x;
    long long s = 0;
    for (int e = threadIdx.x; e < v.N; e += blockDim.x) s += v.metric_env[(size_t)q * v.N + e];
    s = wave_sum_i64(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int i = 0; i < 256 / kWave; ++i) t += part[i];
        v.metric_scalars[q] = t;
    }
}

// metric mode: keyword profit of the step just finished -> running sums.  Outputs are float32 dollars;
// cents are recovered exactly for |x| < 2^22 cents (the division by 100 was one correctly rounded op).
__global__ void k_metric_accumulate(View v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nk = (size_t)v.N * v.K;
    if (i >= nk) return;
    const long long r = (long long)__builtin_rintf(v.rev[i] * 100.0f);
    const long long c = v.model == ADC_MODEL_IMPLICIT ? (long long)__builtin_rintf(v.cost[i] * 100.0f)
                                                      : (long long)__double2ll_rn((double)v.cost[i] * 100.0);
    v.metric_kw[i] += r - c;
}

// metric_profit[k] += sum over a slab of envs of metric_kw[env][k]; grid = (K tiles, env slabs); metric_profit is
// zeroed before the launch
constexpr int kColumnSlabs = 64;
__global__ void k_metric_columns(View v)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= v.K) return;
    const int per = (v.N + kColumnSlabs - 1) / kColumnSlabs;
    const int e0 = blockIdx.y * per, e1 = min(v.N, e0 + per);
    long long s = 0;
    for (int e = e0; e < e1; ++e) s += v.metric_kw[(size_t)e * v.K + k];
    if (e1 > e0) atomicAdd((unsigned long long *)&v.metric_profit[k], (unsigned long long)s);
}

// ---- ideal (max expected) profit of a keyword, adcraft/experiment_utils/experiment_metrics.py:20-61 ------------
// One wavefront per keyword.  n_samples competitor bids (cents) are histogrammed in LDS; for every bid b on
// the grid 1..n_bids cents:  idx = #(samples <= b)  (searchsorted side="right", :30),  IR = idx / n (:32),
// idx' = min(idx, n-1) (:33),  cpc = (sum of the idx'+1 smallest samples) / (idx'+1) (:34-35) - i.e. the mean
// of the samples <= b PLUS the next larger one (the reference's off-by-one, reproduced) - and
// profit(b) = max(vol_mean * IR * bctr * (sctr * rev_mean - cpc), 0) (:51-57); ideal = max_b (:59).
constexpr int kIdealBins = 1024;      // cents 0..1022, last bin = everything above
__global__ __launch_bounds__(kWave) void k_ideal_profit(View v, int n_samples, int n_bids, const double *__restrict__ bid_grid,
                                                         const int32_t *tape_samples, double *ideal_out, double *ir_out, double *cpc_out)
{
    __shared__ unsigned int hist[kIdealBins];      // count per cent value (last bin: everything above)
    __shared__ unsigned int cpre[kIdealBins];      // inclusive prefix of counts
    __shared__ unsigned int spre[kIdealBins];      // inclusive prefix of cents (<= 2^21 * 1022 fits for n <= 2^21)
    __shared__ unsigned int nxt[kIdealBins + 1];   // smallest occupied bin >= i (kIdealBins-1 stands for "above")
    __shared__ unsigned int over_min;
    const int lane = threadIdx.x;
    const int env = blockIdx.x / v.K, k = blockIdx.x - env * v.K;
    for (int i = lane; i < kIdealBins; i += kWave) hist[i] = 0u;
    if (lane == 0) over_min = 0xFFFFFFFFu;
    __syncthreads();
    const float loc = param_at(v, ADC_P_A, env, k), scale = param_at(v, ADC_P_B, env, k);
    const uint64_t key = v.key[env];
    const uint32_t tick = v.tick[env];
    for (int i = lane; i < n_samples; i += kWave) {
        int c;
        if (tape_samples) c = tape_samples[(size_t)blockIdx.x * n_samples + i];
        else {
            const adc::U4 w = adc::draw(key, (uint32_t)(i >> 2), adc::ST_METRIC, (uint32_t)k, tick);
            const uint32_t ws = (i & 3) == 0 ? w.x : (i & 3) == 1 ? w.y : (i & 3) == 2 ? w.z : w.w;
            c = adc::laplace_cents(ws, loc, scale);
        }
        if (c < kIdealBins - 1) atomicAdd(&hist[c], 1u);
        else {
            atomicAdd(&hist[kIdealBins - 1], 1u);
            atomicMin(&over_min, (unsigned int)c);
        }
    }
    __syncthreads();
    // prefix sums over bins: each lane owns 16 consecutive bins, wave scan joins them
    constexpr int kPer = kIdealBins / kWave;
    unsigned int lc = 0, ls = 0;
    for (int i = 0; i < kPer; ++i) {
        const int bin = lane * kPer + i;
        const unsigned int h = bin < kIdealBins - 1 ? hist[bin] : 0u;
        lc += h;
        ls += h * (unsigned int)bin;
    }
    const unsigned int bc = (unsigned int)wave_scan_i32((int)lc) - lc, bs = (unsigned int)wave_scan_i32((int)ls) - ls;
    lc = bc; ls = bs;
    for (int i = 0; i < kPer; ++i) {
        const int bin = lane * kPer + i;
        const unsigned int h = bin < kIdealBins - 1 ? hist[bin] : 0u;
        lc += h;
        ls += h * (unsigned int)bin;
        cpre[bin] = lc;
        spre[bin] = ls;
    }
    // smallest occupied bin at or after i: backward over the lane's bins, then across lanes
    unsigned int first = 0xFFFFFFFFu;
    for (int i = kPer - 1; i >= 0; --i) {
        const int bin = lane * kPer + i;
        if (hist[bin] != 0u) first = (unsigned int)bin;
    }
    unsigned int after = 0xFFFFFFFFu;          // first occupied bin in any later lane
    for (int l = kWave - 1; l > 0; --l) {
        const unsigned int f = (unsigned int)__shfl((int)first, l, 64);
        if (lane < l && f < after) after = f;          // evaluated high to low, so "after" ends as the nearest
    }
    {
        unsigned int run = after;
        for (int i = kPer - 1; i >= 0; --i) {
            const int bin = lane * kPer + i;
            if (hist[bin] != 0u) run = (unsigned int)bin;
            nxt[bin] = run == 0xFFFFFFFFu ? (unsigned int)(kIdealBins - 1) : run;
        }
    }
    if (lane == 0) nxt[kIdealBins] = kIdealBins - 1;
    __syncthreads();
    const double vol_mean = param_at(v, ADC_P_VOL_MEAN, env, k), bctr = param_at(v, ADC_P_BCTR, env, k);
    const double margin = (double)param_at(v, ADC_P_SCTR, env, k) * (double)param_at(v, ADC_P_REV_MEAN, env, k);
    double best = 0.0;
    for (int bi = lane; bi < n_bids; bi += kWave) {
        // the reference compares float dollars: sample c/100.0 <= bid, with bid from np.arange (so e.g. its
        // "0.10" is 0.09999999999999999 and excludes 10-cent samples).  b = largest cent value that passes.
        const double bid = bid_grid[bi];
        int b = (int)__builtin_floor(bid * 100.0 + 0.5);
        if (b > kIdealBins - 2) b = kIdealBins - 2;
        if (b >= 0 && !((double)b / 100.0 <= bid)) b -= 1;
        const unsigned long long idx = b >= 0 ? cpre[b] : 0u;          // searchsorted side="right" (:30)
        const unsigned long long sum = b >= 0 ? spre[b] : 0u;
        const double ir = (double)idx / (double)n_samples;             // :32
        double num, den;
        if (idx >= (unsigned long long)n_samples) { num = (double)sum / 100.0; den = (double)n_samples; }   // idx' = n-1
        else {
            const unsigned int nb = nxt[b + 1];                         // sorted[idx]: the smallest sample above the bid
            num = ((double)sum + (nb < (unsigned int)(kIdealBins - 1) ? (double)nb : (double)over_min)) / 100.0;
            den = (double)(idx + 1);                                    // :33-35 inclusive running mean
        }
        const double cpc = num / den;
        if (ir_out) ir_out[(size_t)blockIdx.x * n_bids + bi] = ir;
        if (cpc_out) cpc_out[(size_t)blockIdx.x * n_bids + bi] = cpc;
        double p = vol_mean * ir * bctr * (margin - cpc);               // :51-57
        p = p > 0.0 ? p : 0.0;
        best = p > best ? p : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if (lane == 0 && ideal_out) ideal_out[blockIdx.x] = best;           // :59
}

struct KeygenTables { adc::QuantileTable t[7]; };

__global__ void k_generate_keywords(View v, KeygenTables tabs, float no_vol_prob, uint32_t serial, const uint8_t *mask)
{
    const int env = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= v.K || (mask && !mask[env])) return;
    float out[8];
    adc::generate_implicit_keyword(v.key[env], (uint32_t)k, serial, tabs.t, no_vol_prob, out);
#pragma unroll
    for (int p = 0; p < ADC_P_COUNT; ++p) param_at(v, p, env, k) = out[p];
    if (k == 0) v.drift_pending[env] = 0;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_reset(View v, const uint8_t *mask, const uint64_t *seeds)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= v.N) return;
    if (mask && !mask[env]) return;
    v.day[env] = 0;
    v.cum_cents[env] = 0;
    v.cum[env] = 0.0;
    if (seeds) {
        v.key[env] = splitmix64(seeds[env]);
        v.tick[env] = 0u;
        v.drift_pending[env] = 0;       // keywords are resampled with a seed (gymnasium_kw_env.py:303)
    }
}

__global__ void k_init_keys(View v, uint64_t seed, int64_t env_id_base)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= v.N) return;
    v.key[env] = splitmix64(seed ^ splitmix64((uint64_t)(env_id_base + env) + 0x5851F42D4C957F2Dull));
}

__global__ void k_sample_actions(View v, float lo, float hi, float budget, float *bids, float *budgets)
{
    const int env = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < v.K) {
        const adc::U4 w = adc::draw(v.key[env], 0u, adc::ST_ACTION, (uint32_t)k, v.tick[env]);
        bids[(size_t)env * v.K + k] = adc::synthetic_bid(w.x, lo, hi);
    }
    if (k == 0) budgets[env] = budget;
}

// flat observation row = sorted-key concatenation of the obs dict (adcraft/gymnasium_kw_utils.py:383-390,
// adcraft/wrappers/flat_array.py:74-80): buyside_clicks[K] | cost[K] | cumulative_profit | days_passed |
// impressions[K] | revenue[K] | sellside_conversions[K], all float32
__global__ void k_flatten_obs(View v)
{
    const int env = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int K = v.K;
    float *row = v.flat_obs + (size_t)env * (5 * K + 2);
    if (k < K) {
        const size_t o = (size_t)env * K + k;
        row[k] = (float)v.clk[o];
        row[K + k] = v.cost[o];
        row[2 * K + 2 + k] = (float)v.imp[o];
        row[3 * K + 2 + k] = v.rev[o];
        row[4 * K + 2 + k] = (float)v.conv[o];
    }
    if (k == 0) {
        row[2 * K] = (float)v.cum_profit[env];
        row[2 * K + 1] = (float)v.day_out[env];
    }
}

__global__ void k_unflatten_actions(View v, const float *flat, float *bids, float *budgets)
{
    // FlatArrayWrapper action = [budget, keyword_bids...] (sorted keys, adcraft/wrappers/flat_array.py:52,76)
    const int env = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const float *row = flat + (size_t)env * (v.K + 1);
    if (k < v.K) bids[(size_t)env * v.K + k] = row[1 + k];
    if (k == 0) budgets[env] = row[0];
}

// nth_price_auction (adcraft/synthetic_kw_helpers.py:116-180): one lane per auction, the top (w+n)
// competitor bids kept sorted ascending in registers.
constexpr int kTopMax = 32;
__global__ void k_nth_price(double bid, const double *__restrict__ other, int n_auctions, int n_bidders, int n,
                            int num_winners, int *__restrict__ won, int *__restrict__ placement, double *__restrict__ cost)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_auctions) return;
    const int top = num_winners + n;
    double best[kTopMax];                            // kept ascending; best[0] = smallest of the top
    int have = 0;
    const double *row = other + (size_t)a * n_bidders;
    const int pad = n_bidders < top ? top - n_bidders : 0;   // not enough bidders: zero bids (:156-161)
    for (int b = -pad; b < n_bidders; ++b) {
        const double x = b < 0 ? 0.0 : row[b];
        if (have < top) {
            int j = have++;
            best[j] = x;
            for (; j > 0 && best[j] < best[j - 1]; --j) { const double t = best[j]; best[j] = best[j - 1]; best[j - 1] = t; }
        } else if (x > best[0]) {
            int j = 0;
            best[0] = x;
            for (; j + 1 < top && best[j] > best[j + 1]; ++j) { const double t = best[j]; best[j] = best[j + 1]; best[j + 1] = t; }
        }
    }
    int index = 0;                                   // searchsorted(auction, bid) side="left" (:167)
    while (index < top && best[index] < bid) ++index;
    const bool w = index > n;                        // :170
    won[a] = w;
    if (w) {
        placement[a] = top - index;                  // :172
        int ci = index - (n - 1);
        if (ci < 0) ci = 0;
        cost[a] = n > 1 ? best[ci] : bid;            // :173-177
    }
}

}  // namespace adck
using namespace adck;

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
struct adc_engine {
    adc_config cfg;
    View v;
    hipStream_t stream = nullptr;
    float *d_bids = nullptr, *d_budget = nullptr;
    std::vector<void *> allocs;
    bool have_reset = false;
    bool profiling = false;
    std::vector<hipEvent_t> ev;          // [kProfileRing][kProfileMarks]
    int ev_used = 0;
    double prof_ms[3] = {0.0, 0.0, 0.0};   // fast pass | exact pass + tail | metric accumulate
    int64_t prof_launches = 0;
    size_t flat_obs_bytes = 0;
    float *d_flat_obs = nullptr;
};

namespace {

template <typename T>
int dev_alloc(adc_engine *e, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t err = hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : sizeof(T));
    if (err != hipSuccess) return fail(ADC_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(err));
    e->allocs.push_back(q);
    err = hipMemsetAsync(q, 0, count * sizeof(T) > 0 ? count * sizeof(T) : sizeof(T), e->stream);
    if (err != hipSuccess) return fail(ADC_EHIP, std::string("hipMemsetAsync: ") + hipGetErrorString(err));
    *p = static_cast<T *>(q);
    return ADC_OK;
}

int flush_profile(adc_engine *e)
{
    for (int i = 0; i < e->ev_used; ++i) {
        hipEvent_t *m = &e->ev[(size_t)i * kProfileMarks];
        HIP_TRY(hipEventSynchronize(m[kProfileMarks - 1]));
        for (int j = 0; j < 3; ++j) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, m[j], m[j + 1]));
            e->prof_ms[j] += ms;
        }
        e->prof_launches += 1;
    }
    e->ev_used = 0;
    return ADC_OK;
}

int launch_step(adc_engine *e, const float *d_bids, const float *d_budget, const TapeView *tape)
{
    View &v = e->v;
    const int N = v.N, K = v.K;
    const bool implicit = v.model == ADC_MODEL_IMPLICIT;
    const size_t lds = (size_t)K * (4 * 4 + 3 * 8);
    if (lds > 160 * 1024) return fail(ADC_EINVAL, "num_keywords too large for the exact pass (LDS)");
    TapeView none{};
    if (tape) {
        if (implicit) hipLaunchKernelGGL((k_step_exact<ADC_MODEL_IMPLICIT, true>), dim3(N), dim3(kWave), lds, e->stream, v, d_bids, d_budget, *tape, 0);
        else hipLaunchKernelGGL((k_step_exact<ADC_MODEL_EXPLICIT, true>), dim3(N), dim3(kWave), lds, e->stream, v, d_bids, d_budget, *tape, 0);
        HIP_TRY(hipGetLastError());
        if (v.flat_obs) {
            hipLaunchKernelGGL(k_flatten_obs, dim3((unsigned)((K + 255) / 256), (unsigned)N), dim3(256), 0, e->stream, v);
            HIP_TRY(hipGetLastError());
        }
        return ADC_OK;
    }
    const bool prof = e->profiling;
    if (prof && e->ev_used == kProfileRing) { int rc = flush_profile(e); if (rc) return rc; }
    hipEvent_t *mark = prof ? &e->ev[(size_t)e->ev_used * kProfileMarks] : nullptr;
    if (prof) HIP_TRY(hipEventRecord(mark[0], e->stream));
    if (implicit) {
        const int tiles = (K + kFastBlock - 1) / kFastBlock;
        hipLaunchKernelGGL(k_step_implicit_fast, dim3((unsigned)N * tiles), dim3(kFastBlock), 0, e->stream, v, d_bids);
        HIP_TRY(hipGetLastError());
        if (prof) HIP_TRY(hipEventRecord(mark[1], e->stream));
        if (K <= kRowsMaxK) {
            const size_t lds_rows = (size_t)K * (4 * 8 + 8 * 4 + 1) + (kRowsBlock / kWave) * kRowsRing * 4 + 16;
            hipLaunchKernelGGL(k_step_exact_rows, dim3(N), dim3(kRowsBlock), lds_rows, e->stream, v, d_bids, d_budget);
        } else {
            hipLaunchKernelGGL((k_step_exact<ADC_MODEL_IMPLICIT, false>), dim3(N), dim3(kWave), lds, e->stream, v, d_bids, d_budget, none, 1);
        }
    } else {
        if (prof) HIP_TRY(hipEventRecord(mark[1], e->stream));
        hipLaunchKernelGGL((k_step_exact<ADC_MODEL_EXPLICIT, false>), dim3(N), dim3(kWave), lds, e->stream, v, d_bids, d_budget, none, 0);
    }
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(mark[2], e->stream));
    if (v.metrics_on && !(implicit && K <= kRowsMaxK)) {      // the IMPLICIT fast / row kernels accumulate in their output phase
        const size_t nk = (size_t)N * K;
        hipLaunchKernelGGL(k_metric_accumulate, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, e->stream, v);
        HIP_TRY(hipGetLastError());
    }
    if (v.flat_obs) {
        hipLaunchKernelGGL(k_flatten_obs, dim3((unsigned)((K + 255) / 256), (unsigned)N), dim3(256), 0, e->stream, v);
        HIP_TRY(hipGetLastError());
    }
    if (prof) { HIP_TRY(hipEventRecord(mark[3], e->stream)); e->ev_used++; }
    return ADC_OK;
}

int fetch(adc_engine *e, adc_step_out *out)
{
    const View &v = e->v;
    const size_t nk = (size_t)v.N * v.K, n = (size_t)v.N;
    if (out) {
        if (out->impressions) HIP_TRY(hipMemcpyAsync(out->impressions, v.imp, nk * 4, hipMemcpyDeviceToHost, e->stream));
        if (out->buyside_clicks) HIP_TRY(hipMemcpyAsync(out->buyside_clicks, v.clk, nk * 4, hipMemcpyDeviceToHost, e->stream));
        if (out->sellside_conversions) HIP_TRY(hipMemcpyAsync(out->sellside_conversions, v.conv, nk * 4, hipMemcpyDeviceToHost, e->stream));
        if (out->cost) HIP_TRY(hipMemcpyAsync(out->cost, v.cost, nk * 4, hipMemcpyDeviceToHost, e->stream));
        if (out->revenue) HIP_TRY(hipMemcpyAsync(out->revenue, v.rev, nk * 4, hipMemcpyDeviceToHost, e->stream));
        if (out->reward) HIP_TRY(hipMemcpyAsync(out->reward, v.reward, n * 8, hipMemcpyDeviceToHost, e->stream));
        if (out->cumulative_profit) HIP_TRY(hipMemcpyAsync(out->cumulative_profit, v.cum_profit, n * 8, hipMemcpyDeviceToHost, e->stream));
        if (out->days_passed) HIP_TRY(hipMemcpyAsync(out->days_passed, v.day_out, n * 4, hipMemcpyDeviceToHost, e->stream));
        if (out->terminated) HIP_TRY(hipMemcpyAsync(out->terminated, v.term, n, hipMemcpyDeviceToHost, e->stream));
        if (out->truncated) HIP_TRY(hipMemcpyAsync(out->truncated, v.trunc, n, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

}  // namespace

ADC_EXPORT int adc_abi_version(void) { return ADC_ABI_VERSION; }
ADC_EXPORT const char *adc_last_error(void) { return g_err.c_str(); }

ADC_EXPORT int adc_device_count(int *count)
{
    if (!count) return fail(ADC_EINVAL, "count is NULL");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) { *count = 0; return fail(ADC_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(err)); }
    *count = n;
    return ADC_OK;
}

ADC_EXPORT void adc_engine_destroy(adc_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto ev : e->ev) (void)hipEventDestroy(ev);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

ADC_EXPORT int adc_engine_create(const adc_config *cfg, adc_engine **out)
{
    if (!cfg || !out) return fail(ADC_EINVAL, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->struct_size != sizeof(adc_config)) return fail(ADC_EINVAL, "adc_config.struct_size mismatch (ABI)");
    if (cfg->num_envs <= 0 || cfg->num_keywords <= 0) return fail(ADC_EINVAL, "num_envs and num_keywords must be positive");
    if (cfg->model != ADC_MODEL_IMPLICIT && cfg->model != ADC_MODEL_EXPLICIT) return fail(ADC_EINVAL, "unknown model");
    if ((int64_t)cfg->num_envs * cfg->num_keywords > ((int64_t)1 << 31) - 1) return fail(ADC_EINVAL, "num_envs*num_keywords exceeds 2^31-1 per device");
    if ((size_t)cfg->num_keywords * 40 > 160 * 1024) return fail(ADC_EINVAL, "num_keywords > 4096 is not supported");
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev <= 0)
        return fail(ADC_EHIP, "no usable HIP device: this engine has no CPU path (hipGetDeviceCount: " +
                                  std::string(err == hipSuccess ? "0 devices" : hipGetErrorString(err)) + ")");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(ADC_EINVAL, "device_id out of range");
    HIP_TRY(hipSetDevice(cfg->device_id));
    adc_engine *e = new (std::nothrow) adc_engine();
    if (!e) return fail(ADC_ENOMEM, "host allocation failed");
    e->cfg = *cfg;
    int rc = ADC_OK;
    auto bail = [&](int code) { adc_engine_destroy(e); return code; };
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(ADC_EHIP, "hipStreamCreate failed"));
    View &v = e->v;
    std::memset(&v, 0, sizeof(v));
    v.N = cfg->num_envs; v.K = cfg->num_keywords; v.model = cfg->model; v.max_days = cfg->max_days;
    v.loss_threshold = cfg->loss_threshold;
    v.drift_vol = cfg->drift_vol; v.drift_ctr = cfg->drift_ctr; v.drift_cvr = cfg->drift_cvr;
    v.drift_on = cfg->drift_enabled ? 1 : 0;
    v.imp_thresh = cfg->impression_thresh;
    v.auto_reset = cfg->auto_reset ? 1 : 0;
    const size_t N = v.N, K = v.K, NK = N * K;
#define A(ptr, count) if ((rc = dev_alloc(e, &(ptr), (count))) != ADC_OK) return bail(rc)
    A(v.params, ADC_P_COUNT * NK);
    A(v.key, N); A(v.tick, N); A(v.day, N); A(v.cum_cents, N); A(v.cum, N); A(v.drift_pending, N); A(v.exact_hint, N);
    A(v.env_cost, N); A(v.env_profit, N);
    A(v.imp, NK); A(v.clk, NK); A(v.conv, NK); A(v.cost, NK); A(v.rev, NK);
    A(v.reward, N); A(v.cum_profit, N); A(v.day_out, N); A(v.term, N); A(v.trunc, N);
    A(v.metric_profit, K); A(v.metric_scalars, 8); A(v.metric_env, 4 * N); A(v.metric_kw, NK);
    A(e->d_bids, NK); A(e->d_budget, N);
#undef A
    hipLaunchKernelGGL(k_build_log_table, dim3(1), dim3(256), 0, e->stream);
    hipLaunchKernelGGL(k_init_keys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, e->stream, v, cfg->seed, cfg->env_id_base);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        return bail(fail(ADC_EHIP, "engine initialisation kernel failed (is this a gfx950 device?)"));
    *out = e;
    return ADC_OK;
}

#define ENGINE_GUARD(e)                                                   \
    if (!(e)) return fail(ADC_EINVAL, "engine handle is NULL");           \
    HIP_TRY(hipSetDevice((e)->cfg.device_id))

ADC_EXPORT int adc_engine_set_params(adc_engine *e, int param_id, const float *host_nk)
{
    ENGINE_GUARD(e);
    if (param_id < 0 || param_id >= ADC_P_COUNT || !host_nk) return fail(ADC_EINVAL, "bad param_id or NULL buffer");
    const size_t NK = (size_t)e->v.N * e->v.K;
    HIP_TRY(hipMemcpyAsync(e->v.params + (size_t)param_id * NK, host_nk, NK * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_get_params(adc_engine *e, int param_id, float *host_nk)
{
    ENGINE_GUARD(e);
    if (param_id < 0 || param_id >= ADC_P_COUNT || !host_nk) return fail(ADC_EINVAL, "bad param_id or NULL buffer");
    const size_t NK = (size_t)e->v.N * e->v.K;
    if (e->v.drift_on) {
        hipLaunchKernelGGL(k_materialize_drift, dim3(e->v.N), dim3(256), 0, e->stream, e->v);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(host_nk, e->v.params + (size_t)param_id * NK, NK * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_set_env_params(adc_engine *e, int env, const float *host_8k)
{
    ENGINE_GUARD(e);
    if (env < 0 || env >= e->v.N || !host_8k) return fail(ADC_EINVAL, "env out of range or NULL buffer");
    const size_t K = e->v.K, NK = (size_t)e->v.N * K;
    for (int p = 0; p < ADC_P_COUNT; ++p)
        HIP_TRY(hipMemcpyAsync(e->v.params + p * NK + env * K, host_8k + p * K, K * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_generate_keywords(adc_engine *e, const adc_quantiles *q, float no_vol_prob, uint32_t serial,
                                            const uint8_t *env_mask)
{
    ENGINE_GUARD(e);
    if (e->v.model != ADC_MODEL_IMPLICIT) return fail(ADC_EINVAL, "device keyword generation is provided for IMPLICIT keywords");
    if (!q) return fail(ADC_EINVAL, "quantile tables are NULL");
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };
    KeygenTables tabs;
    for (int i = 0; i < 7; ++i) {
        const int B = q->buckets[i];
        if (B <= 0 || !q->mins[i] || !q->medians[i] || !q->maxs[i]) { cleanup(); return fail(ADC_EINVAL, "every quantity needs at least one bucket"); }
        float *d = nullptr;
        if (hipMalloc((void **)&d, (size_t)B * 12) != hipSuccess) { cleanup(); return fail(ADC_ENOMEM, "hipMalloc (quantiles) failed"); }
        tmp.push_back(d);
        if (hipMemcpyAsync(d, q->mins[i], (size_t)B * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(d + B, q->medians[i], (size_t)B * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(d + 2 * B, q->maxs[i], (size_t)B * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess) { cleanup(); return fail(ADC_EHIP, "quantile upload failed"); }
        tabs.t[i] = adc::QuantileTable{d, d + B, d + 2 * B, B};
    }
    uint8_t *d_mask = nullptr;
    if (env_mask) {
        if (hipMalloc((void **)&d_mask, (size_t)e->v.N) != hipSuccess) { cleanup(); return fail(ADC_ENOMEM, "hipMalloc (mask) failed"); }
        tmp.push_back(d_mask);
        if (hipMemcpyAsync(d_mask, env_mask, (size_t)e->v.N, hipMemcpyHostToDevice, e->stream) != hipSuccess) { cleanup(); return fail(ADC_EHIP, "mask upload failed"); }
    }
    hipLaunchKernelGGL(k_generate_keywords, dim3((unsigned)((e->v.K + 255) / 256), (unsigned)e->v.N), dim3(256), 0, e->stream, e->v, tabs,
                       no_vol_prob, serial, d_mask);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    cleanup();
    HIP_TRY(err);
    return ADC_OK;
}

ADC_EXPORT int adc_engine_reset(adc_engine *e, const uint8_t *env_mask, const uint64_t *seeds)
{
    ENGINE_GUARD(e);
    const size_t N = e->v.N;
    uint8_t *d_mask = nullptr;
    uint64_t *d_seeds = nullptr;
    if (env_mask) { HIP_TRY(hipMalloc((void **)&d_mask, N)); HIP_TRY(hipMemcpyAsync(d_mask, env_mask, N, hipMemcpyHostToDevice, e->stream)); }
    if (seeds) { HIP_TRY(hipMalloc((void **)&d_seeds, N * 8)); HIP_TRY(hipMemcpyAsync(d_seeds, seeds, N * 8, hipMemcpyHostToDevice, e->stream)); }
    hipLaunchKernelGGL(k_reset, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, e->stream, e->v, d_mask, d_seeds);
    hipError_t err = hipGetLastError();
    hipError_t err2 = hipStreamSynchronize(e->stream);
    if (d_mask) (void)hipFree(d_mask);
    if (d_seeds) (void)hipFree(d_seeds);
    HIP_TRY(err);
    HIP_TRY(err2);
    e->have_reset = true;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_set_limits(adc_engine *e, int32_t max_days, double loss_threshold)
{
    ENGINE_GUARD(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->v.max_days = max_days;
    e->v.loss_threshold = loss_threshold;
    e->cfg.max_days = max_days;
    e->cfg.loss_threshold = loss_threshold;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_set_drift(adc_engine *e, int32_t enabled, float drift_vol, float drift_ctr, float drift_cvr)
{
    ENGINE_GUARD(e);
    if (!enabled && e->v.drift_on) {      // apply what is pending under the old setting, then switch off
        hipLaunchKernelGGL(k_materialize_drift, dim3(e->v.N), dim3(256), 0, e->stream, e->v);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->v.drift_on = enabled ? 1 : 0;
    e->v.drift_vol = drift_vol; e->v.drift_ctr = drift_ctr; e->v.drift_cvr = drift_cvr;
    if (!enabled) HIP_TRY(hipMemsetAsync(e->v.drift_pending, 0, (size_t)e->v.N, e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_get_rng_state(adc_engine *e, uint64_t *keys_n, uint32_t *ticks_n)
{
    ENGINE_GUARD(e);
    const size_t N = e->v.N;
    if (keys_n) HIP_TRY(hipMemcpyAsync(keys_n, e->v.key, N * 8, hipMemcpyDeviceToHost, e->stream));
    if (ticks_n) HIP_TRY(hipMemcpyAsync(ticks_n, e->v.tick, N * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_set_rng_state(adc_engine *e, const uint64_t *keys_n, const uint32_t *ticks_n)
{
    ENGINE_GUARD(e);
    const size_t N = e->v.N;
    if (keys_n) HIP_TRY(hipMemcpyAsync(e->v.key, keys_n, N * 8, hipMemcpyHostToDevice, e->stream));
    if (ticks_n) HIP_TRY(hipMemcpyAsync(e->v.tick, ticks_n, N * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_get_episode_state(adc_engine *e, int32_t *day_n, double *cum_profit_n)
{
    ENGINE_GUARD(e);
    const size_t N = e->v.N;
    if (day_n) HIP_TRY(hipMemcpyAsync(day_n, e->v.day, N * 4, hipMemcpyDeviceToHost, e->stream));
    if (cum_profit_n) {
        if (e->v.model == ADC_MODEL_IMPLICIT) {
            std::vector<int64_t> c(N);
            HIP_TRY(hipMemcpyAsync(c.data(), e->v.cum_cents, N * 8, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            for (size_t i = 0; i < N; ++i) cum_profit_n[i] = (double)c[i] / 100.0;
        } else {
            HIP_TRY(hipMemcpyAsync(cum_profit_n, e->v.cum, N * 8, hipMemcpyDeviceToHost, e->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_set_episode_state(adc_engine *e, const int32_t *day_n, const double *cum_profit_n)
{
    ENGINE_GUARD(e);
    const size_t N = e->v.N;
    if (day_n) HIP_TRY(hipMemcpyAsync(e->v.day, day_n, N * 4, hipMemcpyHostToDevice, e->stream));
    if (cum_profit_n) {
        std::vector<int64_t> c(N);
        for (size_t i = 0; i < N; ++i) c[i] = (int64_t)std::llrint(cum_profit_n[i] * 100.0);
        HIP_TRY(hipMemcpyAsync(e->v.cum_cents, c.data(), N * 8, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->v.cum, cum_profit_n, N * 8, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->have_reset = true;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_step_device(adc_engine *e, const float *d_bids_nk, const float *d_budget_n)
{
    ENGINE_GUARD(e);
    if (!e->have_reset) return fail(ADC_ESTATE, "reset required, need to generate keywords to bid on");
    return launch_step(e, d_bids_nk ? d_bids_nk : e->d_bids, d_budget_n ? d_budget_n : e->d_budget, nullptr);
}

ADC_EXPORT int adc_engine_fetch(adc_engine *e, adc_step_out *out)
{
    ENGINE_GUARD(e);
    return fetch(e, out);
}

ADC_EXPORT int adc_engine_synchronize(adc_engine *e)
{
    ENGINE_GUARD(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_step(adc_engine *e, const float *bids_nk, const float *budget_n, adc_step_out *out)
{
    ENGINE_GUARD(e);
    if (!e->have_reset) return fail(ADC_ESTATE, "reset required, need to generate keywords to bid on");
    if (!bids_nk || !budget_n) return fail(ADC_EINVAL, "bids/budget is NULL");
    const size_t NK = (size_t)e->v.N * e->v.K;
    HIP_TRY(hipMemcpyAsync(e->d_bids, bids_nk, NK * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_budget, budget_n, (size_t)e->v.N * 4, hipMemcpyHostToDevice, e->stream));
    int rc = launch_step(e, e->d_bids, e->d_budget, nullptr);
    if (rc) return rc;
    return fetch(e, out);
}

ADC_EXPORT int adc_engine_step_replay(adc_engine *e, const float *bids_nk, const float *budget_n, const adc_tape *tape,
                                      adc_step_out *out)
{
    ENGINE_GUARD(e);
    if (!e->have_reset) return fail(ADC_ESTATE, "reset required, need to generate keywords to bid on");
    if (!bids_nk || !budget_n || !tape || !tape->volumes) return fail(ADC_EINVAL, "bids/budget/tape is NULL");
    const size_t N = e->v.N, K = e->v.K, NK = N * K;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };
    auto up = [&](const void *host, size_t bytes, const void **dev) -> int {
        *dev = nullptr;
        if (!host) return ADC_OK;
        void *q = nullptr;
        if (hipMalloc(&q, bytes ? bytes : 8) != hipSuccess) return fail(ADC_ENOMEM, "hipMalloc (tape) failed");
        tmp.push_back(q);
        if (bytes && hipMemcpyAsync(q, host, bytes, hipMemcpyHostToDevice, e->stream) != hipSuccess) return fail(ADC_EHIP, "tape upload failed");
        *dev = q;
        return ADC_OK;
    };
    TapeView tv{};
    int rc = ADC_OK;
#define UP(field, type, count) if (!rc) rc = up(tape->field, (size_t)(count) * sizeof(type), (const void **)&tv.field)
    UP(volumes, int32_t, NK);
    UP(bid_cents, int32_t, tape->len_bid);
    UP(x_impressions, int32_t, tape->len_ximp);
    UP(x_cost, double, tape->len_xcost);
    UP(click, uint8_t, tape->len_click);
    UP(conv, uint8_t, tape->len_conv);
    UP(rev_cents, int32_t, tape->len_rev);
    UP(off_bid, int64_t, N); UP(off_ximp, int64_t, N); UP(off_xcost, int64_t, N);
    UP(off_click, int64_t, N); UP(off_conv, int64_t, N); UP(off_rev, int64_t, N);
#undef UP
    int64_t *d_end = nullptr;
    if (!rc && hipMalloc((void **)&d_end, 6 * N * 8) != hipSuccess) rc = fail(ADC_ENOMEM, "hipMalloc (tape cursors) failed");
    if (rc) { cleanup(); return rc; }
    tmp.push_back(d_end);
    tv.end_bid = d_end; tv.end_ximp = d_end + N; tv.end_xcost = d_end + 2 * N;
    tv.end_click = d_end + 3 * N; tv.end_conv = d_end + 4 * N; tv.end_rev = d_end + 5 * N;
    tv.len_bid = tape->len_bid; tv.len_ximp = tape->len_ximp; tv.len_xcost = tape->len_xcost;
    tv.len_click = tape->len_click; tv.len_conv = tape->len_conv; tv.len_rev = tape->len_rev;
    hipError_t err = hipMemcpyAsync(e->d_bids, bids_nk, NK * 4, hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(e->d_budget, budget_n, N * 4, hipMemcpyHostToDevice, e->stream);
    if (err != hipSuccess) { cleanup(); return fail(ADC_EHIP, "action upload failed"); }
    rc = launch_step(e, e->d_bids, e->d_budget, &tv);
    if (!rc) rc = fetch(e, out);
    if (!rc) {
        int64_t *dst[6] = {tape->end_bid, tape->end_ximp, tape->end_xcost, tape->end_click, tape->end_conv, tape->end_rev};
        for (int i = 0; i < 6 && !rc; ++i)
            if (dst[i] && hipMemcpy(dst[i], d_end + i * N, N * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ADC_EHIP, "cursor download failed");
    }
    cleanup();
    return rc;
}

ADC_EXPORT int adc_engine_update_keywords(adc_engine *e)
{
    ENGINE_GUARD(e);
    if (!e->v.drift_on) return ADC_OK;        // updater_mask is None: no-op (gymnasium_kw_env.py:125-126)
    hipLaunchKernelGGL(k_materialize_drift, dim3(e->v.N), dim3(256), 0, e->stream, e->v);
    hipLaunchKernelGGL(k_force_drift, dim3(e->v.N), dim3(256), 0, e->stream, e->v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_host_alloc(size_t bytes, void **out)
{
    if (!out) return fail(ADC_EINVAL, "out is NULL");
    *out = nullptr;
    void *p = nullptr;
    hipError_t err = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (err != hipSuccess) return fail(ADC_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(err));
    *out = p;
    return ADC_OK;
}

ADC_EXPORT void adc_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

ADC_EXPORT int adc_engine_device_buffer(adc_engine *e, int buffer_id, void **dptr, size_t *bytes)
{
    ENGINE_GUARD(e);
    if (!dptr) return fail(ADC_EINVAL, "dptr is NULL");
    const View &v = e->v;
    const size_t N = v.N, K = v.K, NK = N * K;
    void *p = nullptr;
    size_t b = 0;
    switch (buffer_id) {
    case ADC_BUF_PARAMS: p = v.params; b = ADC_P_COUNT * NK * 4; break;
    case ADC_BUF_BIDS: p = e->d_bids; b = NK * 4; break;
    case ADC_BUF_BUDGET: p = e->d_budget; b = N * 4; break;
    case ADC_BUF_IMPRESSIONS: p = v.imp; b = NK * 4; break;
    case ADC_BUF_CLICKS: p = v.clk; b = NK * 4; break;
    case ADC_BUF_CONVERSIONS: p = v.conv; b = NK * 4; break;
    case ADC_BUF_COST: p = v.cost; b = NK * 4; break;
    case ADC_BUF_REVENUE: p = v.rev; b = NK * 4; break;
    case ADC_BUF_REWARD: p = v.reward; b = N * 8; break;
    case ADC_BUF_CUM_PROFIT: p = v.cum_profit; b = N * 8; break;
    case ADC_BUF_DAYS: p = v.day_out; b = N * 4; break;
    case ADC_BUF_TERMINATED: p = v.term; b = N; break;
    case ADC_BUF_TRUNCATED: p = v.trunc; b = N; break;
    case ADC_BUF_METRIC_PROFIT: p = v.metric_profit; b = K * 8; break;
    case ADC_BUF_METRIC_SCALARS: p = v.metric_scalars; b = 8 * 8; break;
    case ADC_BUF_FLAT_OBS:
        if (!v.flat_obs) return fail(ADC_ESTATE, "flat observations are not enabled (adc_engine_flat_obs_enable)");
        p = v.flat_obs; b = N * (5 * K + 2) * 4; break;
    default: return fail(ADC_EINVAL, "unknown buffer id");
    }
    *dptr = p;
    if (bytes) *bytes = b;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_stream(adc_engine *e, void **hip_stream)
{
    ENGINE_GUARD(e);
    if (!hip_stream) return fail(ADC_EINVAL, "hip_stream is NULL");
    *hip_stream = (void *)e->stream;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_sample_actions(adc_engine *e, float bid_lo, float bid_hi, float budget)
{
    ENGINE_GUARD(e);
    hipLaunchKernelGGL(k_sample_actions, dim3((unsigned)((e->v.K + 255) / 256), (unsigned)e->v.N), dim3(256), 0, e->stream, e->v,
                       bid_lo, bid_hi, budget, e->d_bids, e->d_budget);
    HIP_TRY(hipGetLastError());
    return ADC_OK;
}

ADC_EXPORT int adc_engine_set_flat_actions_device(adc_engine *e, const float *d_flat)
{
    ENGINE_GUARD(e);
    if (!d_flat) return fail(ADC_EINVAL, "flat action pointer is NULL");
    hipLaunchKernelGGL(k_unflatten_actions, dim3((unsigned)((e->v.K + 255) / 256), (unsigned)e->v.N), dim3(256), 0, e->stream, e->v,
                       d_flat, e->d_bids, e->d_budget);
    HIP_TRY(hipGetLastError());
    return ADC_OK;
}

ADC_EXPORT int adc_engine_flat_obs_enable(adc_engine *e, int enabled)
{
    ENGINE_GUARD(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (enabled && !e->d_flat_obs) {
        int rc = dev_alloc(e, &e->d_flat_obs, (size_t)e->v.N * (5 * (size_t)e->v.K + 2));
        if (rc) return rc;
    }
    e->v.flat_obs = enabled ? e->d_flat_obs : nullptr;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_profile_enable(adc_engine *e, int enabled)
{
    ENGINE_GUARD(e);
    if (enabled && e->ev.empty()) {
        e->ev.resize((size_t)kProfileRing * kProfileMarks);
        for (auto &ev : e->ev) HIP_TRY(hipEventCreate(&ev));
    }
    e->profiling = enabled != 0;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_profile_read(adc_engine *e, double *kernel_ms_total, int64_t *launches)
{
    ENGINE_GUARD(e);
    int rc = flush_profile(e);
    if (rc) return rc;
    if (kernel_ms_total) for (int j = 0; j < 3; ++j) kernel_ms_total[j] = e->prof_ms[j];
    if (launches) *launches = e->prof_launches;
    e->prof_ms[0] = e->prof_ms[1] = e->prof_ms[2] = 0.0;
    e->prof_launches = 0;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_metrics_enable(adc_engine *e, int enabled)
{
    ENGINE_GUARD(e);
    e->v.metrics_on = enabled ? 1 : 0;
    return ADC_OK;
}

ADC_EXPORT int adc_engine_metrics_reset(adc_engine *e)
{
    ENGINE_GUARD(e);
    HIP_TRY(hipMemsetAsync(e->v.metric_profit, 0, (size_t)e->v.K * 8, e->stream));
    HIP_TRY(hipMemsetAsync(e->v.metric_scalars, 0, 64, e->stream));
    HIP_TRY(hipMemsetAsync(e->v.metric_env, 0, (size_t)e->v.N * 32, e->stream));
    HIP_TRY(hipMemsetAsync(e->v.metric_kw, 0, (size_t)e->v.N * e->v.K * 8, e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_metrics_read(adc_engine *e, int64_t *keyword_profit_cents_k, int64_t *scalars8)
{
    ENGINE_GUARD(e);
    if (keyword_profit_cents_k) {
        HIP_TRY(hipMemsetAsync(e->v.metric_profit, 0, (size_t)e->v.K * 8, e->stream));
        hipLaunchKernelGGL(k_metric_columns, dim3((unsigned)((e->v.K + 255) / 256), kColumnSlabs), dim3(256), 0, e->stream, e->v);
        HIP_TRY(hipGetLastError());
    }
    if (keyword_profit_cents_k) HIP_TRY(hipMemcpyAsync(keyword_profit_cents_k, e->v.metric_profit, (size_t)e->v.K * 8, hipMemcpyDeviceToHost, e->stream));
    if (scalars8) {
        hipLaunchKernelGGL(k_metric_reduce, dim3(4), dim3(256), 0, e->stream, e->v);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(scalars8, e->v.metric_scalars, 64, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ADC_OK;
}

ADC_EXPORT int adc_engine_ideal_profit(adc_engine *e, int n_samples, const double *bid_grid, int n_bids, double *host_nk)
{
    ENGINE_GUARD(e);
    if (e->v.model != ADC_MODEL_IMPLICIT) return fail(ADC_EINVAL, "ideal profit is defined for IMPLICIT keywords");
    if (n_samples <= 0 || !host_nk || !bid_grid || n_bids <= 0) return fail(ADC_EINVAL, "bad arguments");
    const size_t nk = (size_t)e->v.N * e->v.K;
    if (e->v.drift_on) { hipLaunchKernelGGL(k_materialize_drift, dim3(e->v.N), dim3(256), 0, e->stream, e->v); HIP_TRY(hipGetLastError()); }
    double *d_out = nullptr, *d_grid = nullptr;
    HIP_TRY(hipMalloc((void **)&d_out, nk * 8));
    hipError_t err = hipMalloc((void **)&d_grid, (size_t)n_bids * 8);
    if (err == hipSuccess) err = hipMemcpyAsync(d_grid, bid_grid, (size_t)n_bids * 8, hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(k_ideal_profit, dim3((unsigned)nk), dim3(kWave), 0, e->stream, e->v, n_samples, n_bids, d_grid,
                           (const int32_t *)nullptr, d_out, (double *)nullptr, (double *)nullptr);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(host_nk, d_out, nk * 8, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    (void)hipFree(d_out);
    (void)hipFree(d_grid);
    HIP_TRY(err);
    return ADC_OK;
}

// the estimator alone, on caller-supplied samples of ONE keyword (pins the kernel against the reference's
// get_implicit_kw_bid_cpc_impressions; bid grid = 1..n_bids cents)
ADC_EXPORT int adc_bid_curves_from_samples(int device_id, const int32_t *samples_cents, int32_t n_samples, const double *bid_grid,
                                           int32_t n_bids, double *impression_rate_out, double *cpc_out)
{
    if (!samples_cents || n_samples <= 0 || n_bids <= 0 || !bid_grid || !impression_rate_out || !cpc_out)
        return fail(ADC_EINVAL, "bad arguments");
    HIP_TRY(hipSetDevice(device_id));
    int32_t *d_s = nullptr;
    double *d_ir = nullptr, *d_cpc = nullptr, *d_grid = nullptr;
    float *d_p = nullptr;
    uint64_t *d_key = nullptr;
    uint32_t *d_tick = nullptr;
    auto done = [&](int code) { (void)hipFree(d_s); (void)hipFree(d_ir); (void)hipFree(d_cpc); (void)hipFree(d_p); (void)hipFree(d_key); (void)hipFree(d_tick); (void)hipFree(d_grid); return code; };
    if (hipMalloc((void **)&d_s, (size_t)n_samples * 4) != hipSuccess || hipMalloc((void **)&d_ir, (size_t)n_bids * 8) != hipSuccess ||
        hipMalloc((void **)&d_cpc, (size_t)n_bids * 8) != hipSuccess || hipMalloc((void **)&d_p, ADC_P_COUNT * 4) != hipSuccess ||
        hipMalloc((void **)&d_key, 8) != hipSuccess || hipMalloc((void **)&d_tick, 4) != hipSuccess ||
        hipMalloc((void **)&d_grid, (size_t)n_bids * 8) != hipSuccess)
        return done(fail(ADC_ENOMEM, "hipMalloc failed"));
    if (hipMemcpy(d_grid, bid_grid, (size_t)n_bids * 8, hipMemcpyHostToDevice) != hipSuccess) return done(fail(ADC_EHIP, "upload failed"));
    if (hipMemcpy(d_s, samples_cents, (size_t)n_samples * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_p, 0, ADC_P_COUNT * 4) != hipSuccess ||
        hipMemset(d_key, 0, 8) != hipSuccess || hipMemset(d_tick, 0, 4) != hipSuccess)
        return done(fail(ADC_EHIP, "upload failed"));
    View v;
    std::memset(&v, 0, sizeof(v));
    v.N = 1; v.K = 1; v.params = d_p; v.key = d_key; v.tick = d_tick;
    hipLaunchKernelGGL(k_ideal_profit, dim3(1), dim3(kWave), 0, 0, v, n_samples, n_bids, d_grid, d_s, (double *)nullptr, d_ir, d_cpc);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return done(fail(ADC_EHIP, "k_ideal_profit failed"));
    if (hipMemcpy(impression_rate_out, d_ir, (size_t)n_bids * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(cpc_out, d_cpc, (size_t)n_bids * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(ADC_EHIP, "download failed"));
    return done(ADC_OK);
}

ADC_EXPORT int adc_nth_price_auction(int device_id, double bid, const double *other_bids, int32_t n_auctions, int32_t n_bidders,
                                     int32_t n, int32_t num_winners, int32_t *impressions, int32_t *placements, double *costs)
{
    if (!impressions) return fail(ADC_EINVAL, "impressions is NULL");
    *impressions = 0;
    if (n < 1 || num_winners < 1 || n + num_winners > kTopMax) return fail(ADC_EINVAL, "need n >= 1, num_winners >= 1, n + num_winners <= 32");
    if (n_auctions < 0 || n_bidders < 0) return fail(ADC_EINVAL, "negative shape");
    if (n_auctions == 0) return ADC_OK;
    if (n_bidders > 0 && !other_bids) return fail(ADC_EINVAL, "other_bids is NULL");
    HIP_TRY(hipSetDevice(device_id));
    double *d_other = nullptr, *d_cost = nullptr;
    int *d_won = nullptr, *d_place = nullptr;
    const size_t na = n_auctions, nb = n_bidders;
    int rc = ADC_OK;
    auto done = [&](int code) {
        if (d_other) (void)hipFree(d_other);
        if (d_cost) (void)hipFree(d_cost);
        if (d_won) (void)hipFree(d_won);
        if (d_place) (void)hipFree(d_place);
        return code;
    };
    if (hipMalloc((void **)&d_other, (na * nb ? na * nb : 1) * 8) != hipSuccess || hipMalloc((void **)&d_cost, na * 8) != hipSuccess ||
        hipMalloc((void **)&d_won, na * 4) != hipSuccess || hipMalloc((void **)&d_place, na * 4) != hipSuccess)
        return done(fail(ADC_ENOMEM, "hipMalloc failed"));
    if (na * nb && hipMemcpy(d_other, other_bids, na * nb * 8, hipMemcpyHostToDevice) != hipSuccess) return done(fail(ADC_EHIP, "upload failed"));
    hipLaunchKernelGGL(k_nth_price, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, 0, bid, d_other, n_auctions, n_bidders, n, num_winners, d_won, d_place, d_cost);
    if (hipGetLastError() != hipSuccess) return done(fail(ADC_EHIP, "k_nth_price launch failed"));
    std::vector<int> won(na), place(na);
    std::vector<double> cost(na);
    if (hipMemcpy(won.data(), d_won, na * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(place.data(), d_place, na * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(cost.data(), d_cost, na * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(ADC_EHIP, "download failed"));
    int32_t m = 0;
    for (size_t a = 0; a < na; ++a)
        if (won[a]) {
            if (placements) placements[m] = place[a];
            if (costs) costs[m] = cost[a];
            ++m;
        }
    *impressions = m;
    return done(rc);
}
