/*
 * adcraft_engine.h - C ABI of the MI355X-native vectorised BiddingSimulation step engine.
 *
 * This is the drop-in boundary for the reference's per-step hot path.  The reference crosses
 * its native boundary through the pyo3 module `adcraft.rust` (src/lib.rs:14-15,
 * pyproject.toml:38-40) ~10 times per (sub-timestep, keyword) from
 * adcraft/bidding_simulation.py:44-234 and adcraft/gymnasium_kw_env.py:160-269.  A replacement
 * binds ONE call per step for ALL environments instead (adc_engine_step*), plus the scalar
 * entry points that mirror `adcraft.rust` one-to-one for callers that still want them.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch/Python types.
 *   - every function returns ADC_OK (0) or a negative adc_status; adc_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - host buffers are caller-allocated and only borrowed for the duration of the call; device
 *     state is owned by the engine handle; adc_engine_destroy frees it.
 *   - an engine handle is not thread-safe (one caller at a time, like a gym env); different
 *     handles are independent.  ctypes releases the GIL during calls.
 *   - there is NO CPU backend: with no usable HIP device adc_engine_create fails with ADC_EHIP.
 *
 * Layout: all per-keyword arrays are [num_envs][num_keywords], keyword fastest (row-major),
 * parameter planes are [ADC_P_COUNT][num_envs][num_keywords].
 */
#ifndef ADCRAFT_ENGINE_H
#define ADCRAFT_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADC_ABI_VERSION 5
/* revision of the engine's own random stream (which variate lives at which Philox counter): results under a
 * fixed seed - and golden streams recorded from an engine - are comparable only between libraries of the same revision.
 * 2: IMPLICIT / EXPLICIT layout since round 2; 3, 4: IMPLICIT_GENERAL top bids as order statistics, bidder count by inversion;
 * 5: the competitor bids the ideal-profit estimator samples (stage METRIC) by the auction law's own transform of a word (the step's
 *    own streams are those of revision 4) */
#define ADC_STREAM_REVISION 5

typedef enum adc_status {
    ADC_OK = 0,
    ADC_EINVAL = -1,   /* bad argument (-> ValueError / AssertionError in the Python host layer) */
    ADC_EHIP = -2,     /* HIP runtime failure or no device (-> RuntimeError) */
    ADC_ENOMEM = -3,   /* device or host allocation failed (-> MemoryError) */
    ADC_ESTATE = -4,   /* call not valid in the current state, e.g. step before reset */
    ADC_ETYPE = -5,    /* wrong element type for a reducer shim (-> TypeError, see adcraft.rust tests) */
    ADC_ERCCL = -6     /* librccl missing, or an RCCL call of the metric all-reduce failed (-> RuntimeError) */
} adc_status;

/* keyword model: which of the reference's two Keyword subclasses the engine simulates */
typedef enum adc_model {
    ADC_MODEL_IMPLICIT = 0,  /* ImplicitKeyword: literal 2nd-price auction vs one sampled competitor bid
                                (adcraft/synthetic_kw_classes.py:578-646, gymnasium_kw_utils.py:169-195) */
    ADC_MODEL_EXPLICIT = 1,  /* ExplicitKeyword: sigmoid impression rate + Binomial + parametric cost
                                (adcraft/synthetic_kw_classes.py:457-575, gymnasium_kw_utils.py:67-96) */
    ADC_MODEL_IMPLICIT_GENERAL = 2   /* the DEFAULT ImplicitKeyword (not the env's single-competitor form): B ~ Binomial(max_bidders,
                                participation_rate) bidders drawn once per (sub-timestep, keyword) call, raw Laplace(loc, scale) bids,
                                the literal top-(w+n) clearing of nth_price_auction with n = 2: won iff the bid exceeds the (w)-th
                                highest competitor bid, price = the bid just below ours; float64 money
                                (adcraft/synthetic_kw_classes.py:610-686, adcraft/synthetic_kw_helpers.py:116-180).
                                Parameter planes A / B = bid_loc / bid_scale; pool and winners: adc_engine_set_general_model */
} adc_model;

/* parameter planes (float32).  Slots 2,3 depend on the model. */
typedef enum adc_param {
    ADC_P_VOL_MEAN = 0,   /* volume ~ round(max(N(mean, std), 0)), src/lib.rs:314-325 */
    ADC_P_VOL_STD = 1,
    ADC_P_A = 2,          /* IMPLICIT: competitor-bid Laplace loc   | EXPLICIT: impression_bid_intercept */
    ADC_P_B = 3,          /* IMPLICIT: competitor-bid Laplace scale | EXPLICIT: impression_slope */
    ADC_P_BCTR = 4,       /* buyside_ctr */
    ADC_P_SCTR = 5,       /* sellside_paid_ctr */
    ADC_P_REV_MEAN = 6,   /* revenue ~ round2(max(N(mean, std), 0.01)), synthetic_kw_helpers.py:66-70 */
    ADC_P_REV_STD = 7,
    ADC_P_COUNT = 8
} adc_param;

typedef struct adc_config {
    uint32_t struct_size;      /* sizeof(adc_config), for forward compatibility */
    int32_t device_id;         /* HIP device ordinal */
    int32_t num_envs;          /* environments resident on this device */
    int32_t num_keywords;      /* keywords per environment (BiddingSimulation.num_keywords) */
    int32_t model;             /* adc_model */
    int32_t max_days;          /* gymnasium_kw_env.py:61,228 */
    double loss_threshold;     /* dollars; gymnasium_kw_env.py:60,225 */
    float drift_vol;           /* updater_params [["vol",a],["ctr",b],["cvr",c]], gymnasium_kw_env.py:62 */
    float drift_ctr;
    float drift_cvr;
    int32_t drift_enabled;     /* updater_mask == [True]*K (the only mask the reference's configs use) */
    float impression_thresh;   /* EXPLICIT: impression_thresh, 0.05 in the env (gymnasium_kw_utils.py:81) */
    int32_t auto_reset;        /* vector form: a finished env restarts (day=0, cum=0) after reporting */
    int64_t env_id_base;       /* global id of local env 0 (multi-GPU sharding; used for default keys) */
    uint64_t seed;             /* engine seed; env e gets key = mix(seed, env_id_base + e) until reset with a seed */
} adc_config;

/* caller-allocated host outputs of one step; any pointer may be NULL to skip that copy */
typedef struct adc_step_out {
    int32_t *impressions;      /* [N*K] obs["impressions"] */
    int32_t *buyside_clicks;   /* [N*K] obs["buyside_clicks"] */
    int32_t *sellside_conversions; /* [N*K] */
    float *cost;               /* [N*K] dollars, obs["cost"] */
    float *revenue;            /* [N*K] dollars, obs["revenue"] */
    double *reward;            /* [N]   step profit, gymnasium_kw_env.py:222,230 */
    double *cumulative_profit; /* [N]   after the step, :223,242 */
    int32_t *days_passed;      /* [N]   after the step, :227,243 */
    uint8_t *terminated;       /* [N]   :228 */
    uint8_t *truncated;        /* [N]   :225 */
    /* optional compact form of the three counts: uint16 [N][3][K] = per env: impressions | buyside_clicks |
     * sellside_conversions (K even), packed on the device: 6 B instead of 12 B per keyword over PCIe (a host step of
     * 4096 x 256 is PCIe-bound).  A count above 65535 is stored as 65535 and *counts_overflow (nullable) is set to 1:
     * use the int32 pointers then. */
    uint16_t *counts_u16;
    int32_t *counts_overflow;
} adc_step_out;

/* device-resident buffers of the engine (for zero-copy consumers: torch / DLPack / RL on GPU) */
typedef enum adc_buffer {
    ADC_BUF_PARAMS = 0,        /* float [8][N][K] */
    ADC_BUF_BIDS = 1,          /* float [N][K]  engine-owned action staging buffer */
    ADC_BUF_BUDGET = 2,        /* float [N] */
    ADC_BUF_IMPRESSIONS = 3,   /* int32 [N][K] */
    ADC_BUF_CLICKS = 4,
    ADC_BUF_CONVERSIONS = 5,
    ADC_BUF_COST = 6,          /* float [N][K] */
    ADC_BUF_REVENUE = 7,
    ADC_BUF_REWARD = 8,        /* double [N] */
    ADC_BUF_CUM_PROFIT = 9,    /* double [N] */
    ADC_BUF_DAYS = 10,         /* int32 [N] */
    ADC_BUF_TERMINATED = 11,   /* uint8 [N] */
    ADC_BUF_TRUNCATED = 12,
    ADC_BUF_METRIC_PROFIT = 13,/* int64 [K]  sum over local envs and steps of keyword profit, cents (valid after metrics_read) */
    ADC_BUF_METRIC_SCALARS = 14,/* int64 [8]  {profit_cents, env_steps, episodes, truncations, auctions, 0,0,0} */
    ADC_BUF_FLAT_OBS = 15      /* float [N][5K+2] FlatArrayWrapper layout (adcraft/wrappers/flat_array.py:74-80) */
} adc_buffer;

/* replay ("tape") variate source: the variates the reference drew, in the order it drew them
 * (t-major, keyword-minor; adcraft/bidding_simulation.py:216-233).  Used for bit-exact parity
 * against fixtures recorded from the reference.  Per-env start offsets index the flat tapes;
 * `*_end` (nullable, [N]) receives the cursor after the step. */
typedef struct adc_tape {
    const int32_t *volumes;        /* [N*K] auction volume of each keyword this step */
    const int32_t *bid_cents;      /* IMPLICIT: competitor bids in cents, n per visited cell */
    const int32_t *x_impressions;  /* EXPLICIT: Binomial result per visited cell; IMPLICIT_GENERAL: bidders of the cell */
    const double *x_cost;          /* EXPLICIT: per-impression costs; IMPLICIT_GENERAL: bids, bidders x auctions per cell (bidder-major) */
    const uint8_t *click;          /* one per won auction (IMPLICIT) / per cost entry incl. phantom (EXPLICIT) */
    const uint8_t *conv;           /* one per paid click */
    const int32_t *rev_cents;      /* one per conversion */
    int64_t len_bid, len_ximp, len_xcost, len_click, len_conv, len_rev;  /* tape lengths (bounds checks) */
    const int64_t *off_bid, *off_ximp, *off_xcost, *off_click, *off_conv, *off_rev;   /* [N] start cursors */
    int64_t *end_bid, *end_ximp, *end_xcost, *end_click, *end_conv, *end_rev;         /* [N] nullable */
    /* nullable [3][N*K]: the three coefficient vectors update_keywords() drew, np_random.uniform(-a, a, size=K) in its
     * order vol, ctr, cvr (adcraft/gymnasium_kw_env.py:132-135).  When given (drift must be enabled) the replayed step ends
     * with update_keywords() on exactly these coefficients (:246), applied at once instead of from the engine's stream. */
    const float *drift_uniforms;
} adc_tape;

/* quantile tables for device-side keyword generation: the rows of the reference's quantile DataFrame, per quantity
 * (order: vol, ave_cpc, std_cpc, bctr, sctr, rpsc, std_rpsc), already filtered to count_<param> > 0
 * (adcraft/gymnasium_kw_utils.py:296-332) */
typedef struct adc_quantiles {
    int32_t buckets[7];
    const float *mins[7], *medians[7], *maxs[7];     /* [buckets[i]] each */
} adc_quantiles;

typedef struct adc_engine adc_engine;

/* ---- lifecycle ---------------------------------------------------------------------------------- */
int adc_abi_version(void);
int adc_stream_revision(void);
const char *adc_last_error(void);
int adc_device_count(int *count);
int adc_engine_create(const adc_config *cfg, adc_engine **out);
void adc_engine_destroy(adc_engine *e);

/* ---- keyword state (what reset() generates host-side: gymnasium_kw_env.py:303-316) --------------- */
/* one parameter plane for all envs, host float [N*K] */
int adc_engine_set_params(adc_engine *e, int param_id, const float *host_nk);
int adc_engine_get_params(adc_engine *e, int param_id, float *host_nk);   /* applies pending drift first */
/* all 8 planes of ONE env, host float [8][K] */
int adc_engine_set_env_params(adc_engine *e, int env, const float *host_8k);

/* draw the keyword set of every env with env_mask[e]!=0 (NULL = all) on the device: the law of
 * sample_implicit_keywords_from_quantile_dfs (gymnasium_kw_utils.py:295-339: bucket pick + piecewise-linear
 * interpolation, no_vol_prob, std un-normalisation), from the env's own Philox key (call after a reset with seeds);
 * `serial` distinguishes successive generations under the same key (0 right after a seeded reset, so that the same
 * seed reproduces the same keyword set).
 * Same law as the host recipe, NOT the same numbers as the reference's PCG64 draws (those are reproduced by
 * generating host-side and uploading with adc_engine_set_params). */
int adc_engine_generate_keywords(adc_engine *e, const adc_quantiles *q, float no_vol_prob, uint32_t serial,
                                 const uint8_t *env_mask);

/* the same for the EXPLICIT model (the default constructor's keyword set): the law of sample_random_keywords
 * (gymnasium_kw_utils.py:113-156, draws :129-140) - vol_mean = int(2^Beta(2,5) 15 - 1), vol_std = U 0.5 (vol_mean + 1),
 * sctr ~ Beta(5,2), intercept = 1.5 U, rev_mean = 1.5 Beta(2,5), rev_std = Beta(2,5) rev_mean, bctr ~ Beta(2,5),
 * slope = 25 Beta(5,5) - from the env's own Philox key; every Beta is an order statistic of uniforms (integer parameters: exact).
 * Same law, not the reference's PCG64 numbers (those: sample host-side, adc_engine_set_params). */
int adc_engine_generate_explicit_keywords(adc_engine *e, uint32_t serial, const uint8_t *env_mask);

/* reset(): day=0, cumulative_profit=0 for envs with env_mask[e]!=0 (NULL = all);
 * seeds (nullable, [N]) re-key the env's random stream (reset(seed=...)); gymnasium_kw_env.py:271-346 */
int adc_engine_reset(adc_engine *e, const uint8_t *env_mask, const uint64_t *seeds);

/* reset(options={"max_days":..., "loss_threshold":...}) and set_updater_mask()/updater_params changes
 * (gymnasium_kw_env.py:105-112,318-325) */
int adc_engine_set_limits(adc_engine *e, int32_t max_days, double loss_threshold);
int adc_engine_set_drift(adc_engine *e, int32_t enabled, float drift_vol, float drift_ctr, float drift_cvr);

/* random-stream state of every env: Philox key [N] and step counter ("tick") [N]; with the episode state
 * below this is everything needed to checkpoint / resume an engine (the parameters come from get_params) */
int adc_engine_get_rng_state(adc_engine *e, uint64_t *keys_n, uint32_t *ticks_n);
int adc_engine_set_rng_state(adc_engine *e, const uint64_t *keys_n, const uint32_t *ticks_n);
/* episode state: current_day [N], cumulative_profit in dollars [N] (gymnasium_kw_env.py:327-328) */
int adc_engine_get_episode_state(adc_engine *e, int32_t *day_n, double *cum_profit_n);
int adc_engine_set_episode_state(adc_engine *e, const int32_t *day_n, const double *cum_profit_n);

/* ---- the hot path: BiddingSimulation.step for all envs (gymnasium_kw_env.py:160-269) ------------- */
/* host in / host out, synchronous.  bids [N*K] (action["keyword_bids"]), budget [N] (action["budget"]). */
int adc_engine_step(adc_engine *e, const float *bids_nk, const float *budget_n, adc_step_out *out);
/* device in / device out, asynchronous on the engine's stream (NULL = use the engine's staging buffers) */
int adc_engine_step_device(adc_engine *e, const float *d_bids_nk, const float *d_budget_n);
/* copy the last step's outputs to host buffers (synchronises) */
int adc_engine_fetch(adc_engine *e, adc_step_out *out);
/* Byte offsets of the ten adc_step_out arrays (in the struct's order) inside the engine's device output block, and the
 * block's size.  Host buffers placed at these offsets of one allocation are filled by ONE transfer; equal-sized outputs at
 * a constant host stride (e.g. planes of one [5][N][K] array) by one 2-D transfer; anything else by one transfer each. */
int adc_engine_out_offsets(const adc_engine *e, size_t offsets[10], size_t *block_bytes);
int adc_engine_synchronize(adc_engine *e);
/* the same step with FlatArrayWrapper-layout host I/O (adcraft/wrappers/flat_array.py:44-87), synchronous:
 * flat_actions [N][K+1] = [budget, bids...] in; flat_obs [N][5K+2] (sorted-key order, see
 * adc_engine_flat_obs_enable) out; reward [N], terminated [N], truncated [N] out (nullable).  The un/flattening
 * happens on the device, so the host moves one array each way. */
int adc_engine_step_flat(adc_engine *e, const float *flat_actions, float *flat_obs, double *reward, uint8_t *terminated,
                         uint8_t *truncated);
/* the asynchronous forms: enqueue (actions up, kernels, outputs down) on the engine's stream and return; adc_engine_wait
 * completes them.  Give them page-locked buffers (adc_host_alloc) and the transfers of one engine overlap the kernels of
 * another on the same device: a vector env split over a few engines hides most of its PCIe time that way. */
int adc_engine_step_async(adc_engine *e, const float *bids_nk, const float *budget_n, adc_step_out *out);
int adc_engine_step_flat_async(adc_engine *e, const float *flat_actions_n_k1, float *flat_obs_n_5k2, double *reward_n,
                               uint8_t *terminated_n, uint8_t *truncated_n);
int adc_engine_wait(adc_engine *e);
/* replay a recorded tape instead of the engine's own random stream (parity mode) */
int adc_engine_step_replay(adc_engine *e, const float *bids_nk, const float *budget_n, const adc_tape *tape,
                           adc_step_out *out);
/* ADC_MODEL_IMPLICIT_GENERAL: the bidder pool (ImplicitKeyword._bidder_distribution_init defaults 30, 0.6) and the number of
 * winning placements (ImplicitKeyword.auction's n_winners, default 1); max_bidders in [0, 252], num_winners in {1, 2} */
int adc_engine_set_general_model(adc_engine *e, int32_t max_bidders, float participation_rate, int32_t num_winners);
/* BiddingSimulation.update_keywords() called directly (gymnasium_kw_env.py:114-158) */
int adc_engine_update_keywords(adc_engine *e);

/* page-locked host memory for step I/O buffers (DMA straight to/from the caller's arrays instead of staged
 * pageable copies); plain malloc-style ownership: free with adc_host_free */
int adc_host_alloc(size_t bytes, void **out);
void adc_host_free(void *p);

/* ---- device-resident access ---------------------------------------------------------------------- */
int adc_engine_device_buffer(adc_engine *e, int buffer_id, void **dptr, size_t *bytes);
int adc_engine_stream(adc_engine *e, void **hip_stream);
/* fill the engine's action staging buffers with synthetic actions: bid = round2(U(lo,hi)) from the
 * engine's ACTION stream at the current tick, budget = `budget` for every env */
int adc_engine_sample_actions(adc_engine *e, float bid_lo, float bid_hi, float budget);
/* emit FlatArrayWrapper-compatible observations [N][5K+2] float32 on the device after every step
 * (sorted-key order: buyside_clicks, cost, cumulative_profit, days_passed, impressions, revenue,
 * sellside_conversions; adcraft/wrappers/flat_array.py:74-80).  Read it through ADC_BUF_FLAT_OBS. */
int adc_engine_flat_obs_enable(adc_engine *e, int enabled);
/* FlatArrayWrapper-compatible action [N][K+1] = [budget, bids...] (device pointer) -> staging buffers */
int adc_engine_set_flat_actions_device(adc_engine *e, const float *d_flat_n_k1);

/* ---- measurement --------------------------------------------------------------------------------- */
/* when enabled, the kernels of every step are bracketed by HIP events on the engine stream */
int adc_engine_profile_enable(adc_engine *e, int enabled);
/* bracket only every `every`-th step (default 1: all).  Recording four events a step keeps a step's small kernels from
 * overlapping the next step's launch - about 16 us per step at 0.21 ms; a sampled measurement leaves the throughput alone. */
int adc_engine_profile_sample_every(adc_engine *e, int32_t every);
/* kernel_ms_total[3] = summed durations of {fast pass, exact pass + step tail, metric accumulate} over `launches`
 * MEASURED steps since enable / the last read; resets the counters */
int adc_engine_profile_read(adc_engine *e, double *kernel_ms_total, int64_t *launches);
/* hipEventRecord calls the engine has issued since it was created (four per bracketed step): lets a benchmark show that
 * its timed region recorded none */
int adc_engine_profile_records(adc_engine *e, int64_t *event_records);
/* GPU time of a whole region of the engine's stream from ONE event pair (nothing per step): `begin` records an event, `end`
 * records another, waits for it and returns the milliseconds between the two - the figure a host-clock timing of K steps is
 * checked against */
int adc_engine_region_begin(adc_engine *e);
int adc_engine_region_end(adc_engine *e, double *gpu_ms);
/* how many ENV GROUPS the last step ran as (1: all envs as one launch per kernel on the engine's stream).  A CHAIN of device-resident
 * steps - adc_engine_step_device following adc_engine_step_device, with nothing between them but the device-side calls that touch
 * every env on its own: adc_engine_agent_step, adc_engine_ideal_step without host outputs, adc_engine_policy_oracle,
 * adc_engine_sample_actions; adc_engine_run_days is such a chain - of an engine with 2048 envs or more (up to 1024 keywords; not the
 * default ImplicitKeyword beyond 512 keywords, nor a budget-free IMPLICIT batch of 16 rounds of workgroups or more, which measured
 * slower) runs as 4 contiguous env groups (2 for sparse IMPLICIT keyword sets), each with its own view of the engine's arrays, its
 * own lists and its own HIP stream: the tail of one group's launch and its small latency-bound kernels (step tail, budget-exact
 * kernels, the per-step ideal) run under another group's keyword-parallel pass, and a group starts its next day while another
 * finishes this one.  Scheduling only - results never depend on it.  Any other call ends the chain: it first makes the engine's
 * stream wait for the groups, so callers order their work behind a step exactly as before, and the step behind it runs as one group
 * (forking and joining the groups costs more than the overlap inside a single step returns).  After adc_engine_stream has handed the
 * stream out, and while profiling brackets kernels with events, every step is one group.  The first grouped step of an engine picks
 * the groups' streams - one per hardware queue, found by a 150 us spin kernel on pairs of candidate streams, because two groups on
 * one queue would run one after the other: 15 to 50 ms, once. */
int adc_engine_env_groups(adc_engine *e, int32_t *groups);
/* ... and fixes it: 0 = the engine chooses (the default), 1..4 = that many groups (capped by the env count; 1 is the one-stream
 * schedule of earlier ABI versions).  ADCRAFT_STREAM_GROUPS in the environment sets the same thing at creation. */
int adc_engine_set_env_groups(adc_engine *e, int32_t groups);
/* name of the kernel the last step's first pass ran (the one kernel_ms_total[0] times).  IMPLICIT: "k_step_implicit_fast<false>"
 * (dense keyword sets, 256 keywords per workgroup), "k_step_implicit_fast<true>" (a handful of envs: narrow tiles), either with
 * ", lists" before the ">" once an env lists its clicked wins for k_step_click_walk ("k_step_implicit_fast<false, lists>"),
 * "k_step_implicit_sparse" (few auctions per keyword); IMPLICIT_GENERAL: "k_step_general_fast", "k_step_general_small" (a handful
 * of envs: a wavefront per keyword); EXPLICIT: "k_step_explicit_fast"; "k_step_exact" after a tape replay; "" before the first
 * step.  A static string: do not free. */
const char *adc_engine_step_kernel_name(adc_engine *e);

/* ---- multi-GPU: the one collective of the path ----------------------------------------------- */
/* Envs shard over the GPUs of a node, one process (one engine) per GPU; nothing on the step path communicates.  The
 * episode-level metric (compute_AKNCP / compute_NCP, adcraft/experiment_utils/experiment_metrics.py:64-83) needs sums over
 * ALL envs: one RCCL all-reduce (sum) of 3K + 8 doubles on the engine's stream, over xGMI.
 *   rank 0:      adc_comm_get_unique_id(id)  and hands the 128 bytes to the other ranks (file, socket, ... - caller's choice)
 *   every rank:  adc_engine_comm_init(engine, id, rank, world_size)         (collective)
 *   per report:  adc_engine_metrics_allreduce(engine, ...)                  (collective)
 * Without a communicator (a single GPU) the calls return the local sums.  librccl.so is opened on first use. */
#define ADC_COMM_ID_BYTES 128
int adc_comm_get_unique_id(uint8_t *id_bytes /* [ADC_COMM_ID_BYTES] */);
int adc_engine_comm_init(adc_engine *e, const uint8_t *id_bytes, int32_t rank, int32_t world_size);
int adc_engine_comm_destroy(adc_engine *e);
int adc_engine_comm_info(adc_engine *e, int32_t *rank, int32_t *world_size);
/* out_3k8[0..K) = keyword profit in cents, [K..2K) = ideal profit, [2K..3K) = ideal profit with <= 0 -> 1 per entry (the
 * denominator compute_AKNCP uses, :71-75), [3K..3K+8) = {profit_cents, env_steps, episodes, truncations, ...}: sums over
 * steps, envs and ranks.  The ideal sums are the engine's own (adc_engine_ideal_step) when it accumulates them; otherwise
 * this rank's contribution may be passed as host vectors ideal_k / ideal_pos_k ([K] doubles each, NULL = zeros). */
int adc_engine_metrics_allreduce(adc_engine *e, const double *ideal_k, const double *ideal_pos_k, double *out_3k8);
/* what the metric reductions have cost on the device so far: calls of adc_engine_metrics_allreduce and the milliseconds (HIP events
 * on the engine's stream, three records per call) of this rank's own reduction kernels and of the ncclAllReduce (0 without a
 * communicator); reset != 0 zeroes the counters afterwards.  Any pointer may be NULL. */
int adc_engine_comm_stats(adc_engine *e, int64_t *calls, double *ms_local_reduction, double *ms_allreduce, int reset);
/* sum (op 0) or max (op 1) of `count` host doubles over the ranks, in place (a barrier is count = 1) */
int adc_engine_comm_allreduce_f64(adc_engine *e, double *inout, int32_t count, int32_t op);

/* ---- episode metrics (adcraft/experiment_utils/experiment_metrics.py:64-83) ----------------------- */
int adc_engine_metrics_enable(adc_engine *e, int enabled);
int adc_engine_metrics_reset(adc_engine *e);
/* local (this device) accumulators to host: keyword_profit_cents [K], scalars [8] */
int adc_engine_metrics_read(adc_engine *e, int64_t *keyword_profit_cents_k, int64_t *scalars8);
/* ideal (max expected) profit per keyword from the CURRENT parameters, n_samples sampled competitor bids and an
 * ascending bid grid in dollars (the notebooks use np.arange(0.01, 3.00, 0.01)); experiment_metrics.py:20-61;
 * host double [N*K] */
int adc_engine_ideal_profit(adc_engine *e, int n_samples, const double *bid_grid, int n_bids, double *host_nk);
/* the estimator alone, on caller-supplied competitor-bid samples (cents) of one keyword: impression rate and
 * expected cpc on the given bid grid, exactly as get_implicit_kw_bid_cpc_impressions computes them
 * (experiment_metrics.py:28-37, including its inclusive running-mean index) */
int adc_bid_curves_from_samples(int device_id, const int32_t *samples_cents, int32_t n_samples, const double *bid_grid,
                                int32_t n_bids, double *impression_rate_out, double *cpc_out);

/* ---- the callers of the step, device-resident: per-step ideal profit and the paper's baseline bidders ----------
 * (the loop of adcraft/baseline_experiment_and_figs_notebooks/run_heatmap_experiments.ipynb cell 1 and
 * timing_and_other_one_off_experiments.ipynb cell 2: agent.update_all_caches -> agent.sample_action ->
 * get_max_expected_bid_profits per keyword -> env.step -> profits).  All of it runs on the engine's stream against the
 * engine's device-resident observation and action buffers: no host round trip per step. */

/* impression-rate / expected-cpc curves of every keyword on `bid_grid` from n_samples sampled competitor bids
 * (get_implicit_kw_bid_cpc_impressions, experiment_metrics.py:20-37; the notebooks build them once after reset()).
 * Kept on the device as integer numerators, 8 bytes x N x K x n_bids; n_samples <= 2^20. */
int adc_engine_bid_curves_build(adc_engine *e, int n_samples, const double *bid_grid, int n_bids);
/* the cached curves to host: impression rate and expected cpc, double [N*K][n_bids] each (either may be NULL) */
int adc_engine_bid_curves_fetch(adc_engine *e, double *impression_rate_host, double *cpc_host);
/* get_max_expected_bid_profits (experiment_metrics.py:40-61) for the CURRENT (drifted) parameters against the cached
 * curves: max expected profit and its argmax over the grid, per keyword; host outputs may be NULL.  With metrics
 * enabled the value is also added to the per-keyword ideal sums (raw, and with <= 0 replaced by 1 as compute_AKNCP
 * does, :71-75). */
/* diagnostic: per keyword, the grid points that can be the argmax of the expected profit for SOME margin sctr x rev_mean, each
 * with the margin interval on which it can (found once per adc_engine_bid_curves_build; adc_engine_ideal_step evaluates only
 * the few whose interval holds the day's margin): n_nk[N*K] (0xFFFF: the whole grid is evaluated), entries_nkc6[N*K][*cap][6]
 * = {interval lo, hi (float32 bits), curve point (2 words), grid index, 0}, ascending grid indices */
int adc_engine_bid_curves_contenders(adc_engine *e, uint16_t *n_nk, uint32_t *entries_nkc6, int32_t *cap);
int adc_engine_ideal_step(adc_engine *e, double *ideal_host_nk, int32_t *best_index_host_nk);
/* run_oracle_agent: next action := bid_grid[argmax] of the last adc_engine_ideal_step, budget as given */
int adc_engine_policy_oracle(adc_engine *e, float budget);

/* NaiveZeroMarginStrategy (adcraft/baselines/interpolated_expectations.py:442-515), one agent per env.
 * init: empty caches (:286-295), max_bids = 0.01 (:481), the agent's Philox stream keyed by seeds_n (NULL: derived
 * from adc_config.seed and the env id). */
int adc_engine_agent_init(adc_engine *e, float default_expected_revenue_per_conversion, const uint64_t *seeds_n);
/* update_all_caches (:485-494) with one observation per keyword: host arrays [N*K] (all three), or all NULL = the
 * engine's last observation where it lies on the device (after reset: zeros, as the reference's reset observation) */
int adc_engine_agent_update(adc_engine *e, const int32_t *clicks_nk, const int32_t *conversions_nk, const float *revenue_nk);
/* sample_action (:496-515) into the engine's action buffers (what adc_engine_step_device(e, NULL, NULL) consumes).
 * budget_override > 0 replaces the agent's 100 x sum(codes) budget (the notebooks pass budget = 100000 to the env).
 * replay_uniforms_nk (host double [N*K], nullable): the rng.random() value to use for each keyword instead of the
 * agent's Philox stream - parity mode against recorded reference runs. */
int adc_engine_agent_act(adc_engine *e, float budget_override, const double *replay_uniforms_nk);
/* update from the device-resident observation + act, one launch (the closed loop's per-step call) */
int adc_engine_agent_step(adc_engine *e, float budget_override);
/* caches to host (any pointer may be NULL): ave_rpc, num_rpc_obs, ave_sctr, num_sctr_obs, max_bids, each [N*K] */
int adc_engine_agent_state(adc_engine *e, float *ave_rpc_nk, int32_t *num_rpc_obs_nk, float *ave_sctr_nk,
                           int32_t *num_sctr_obs_nk, double *max_bids_nk);
/* the engine's device action buffers to host (what a policy above, adc_engine_sample_actions or
 * adc_engine_set_flat_actions_device last wrote) */
int adc_engine_get_actions(adc_engine *e, float *bids_nk, float *budget_n);
/* `days` consecutive days of the device-resident loop in one call: per day, the policy writes the action
 * (FIXED_ACTIONS: whatever the action buffers hold; ZERO_MARGIN: adc_engine_agent_step, plus adc_engine_ideal_step when
 * curves are built; ORACLE: adc_engine_ideal_step + adc_engine_policy_oracle), then the env steps.  Asynchronous on
 * the engine's stream.  adc_engine_day_graph_enable(e, 1) makes it replay pairs of days from a captured hipGraph
 * (same results; measured no faster on MI355X - the dependent kernels of a day are latency-, not launch-bound - and an engine
 * whose chain of days runs as env groups, see adc_engine_env_groups, keeps the plain chain, which is the faster of the two). */
enum adc_policy { ADC_POLICY_FIXED_ACTIONS = 0, ADC_POLICY_ZERO_MARGIN = 1, ADC_POLICY_ORACLE = 2 };
int adc_engine_run_days(adc_engine *e, int policy, int32_t days, float budget);
int adc_engine_day_graph_enable(adc_engine *e, int enabled);
/* per (env, keyword) metric sums to host (any pointer may be NULL): profit in cents, ideal, ideal with <= 0 -> 1;
 * per-env AKNCP = median_k(profit / ideal_pos), NCP = sum profit / sum ideal (experiment_metrics.py:64-83) */
int adc_engine_metrics_read_nk(adc_engine *e, int64_t *profit_cents_nk, double *ideal_sum_nk, double *ideal_pos_sum_nk);
/* the same two metrics reduced on the device: akncp_n[N] = median over the keywords of (profit / days) / (ideal_pos / days) (a
 * bitonic sort per env in LDS; numpy's median convention), ncp_n[N]; 2 N doubles cross the bus instead of 3 N K.
 * num_keywords <= 4096. */
int adc_engine_metrics_akncp_ncp(adc_engine *e, double days, double *akncp_n, double *ncp_n);

/* ---- info["bidding_outcomes"] on demand (src/lib.rs:251-275, adcraft/gymnasium_kw_env.py:247-251) -------------------- */
/* The fused step kernels keep per-keyword totals, not the per-click lists the reference formats ('costs', 'revenues',
 * 'revenues_per_cost').  Every variate is addressed by (env key; index, stage, keyword, tick), so those lists can be
 * regenerated exactly, only when somebody reads them: this call walks one env-step once more, read-only, in the reference's
 * order (sub-timestep, keyword, auction; the budget walk in its floating point) and lists the paid clicks: keyword[i],
 * timestep[i], cost[i] (dollars), revenue[i] (dollars; -1 = the click did not convert).  steps_back = 1: env `env`'s LAST
 * step; k > 1: the k-th last - valid while no reset / tape replay lies in between, drift is off and the caller has not
 * changed the env's parameters since (the stream position is recomputed, the parameters are read as they stand).  *count =
 * paid clicks of the step (may exceed capacity: then only `capacity` are stored).  share_volume_k[K] (optional) = per
 * keyword, the auctions of the sub-timesteps that had an impression - the denominator combine_outcomes ends up with for
 * 'impression_share' (bidding_simulation.py:130-146).  `bids_k` and `budget` are that step's action for this env.
 * ADC_ESTATE when the step cannot be replayed. */
int adc_engine_outcomes_replay(adc_engine *e, int32_t env, int32_t steps_back, const float *bids_k, float budget, int64_t capacity,
                               int32_t *keyword, int32_t *timestep, double *cost, double *revenue, int64_t *count, int32_t *share_volume_k);
/* The same lists for a step given as a tape (parity mode: the variates the reference drew, adc_engine_step_replay): env `env`
 * is walked over `tape` (its volumes, per-env start offsets and lengths as for adc_engine_step_replay; the end cursors are not
 * written), read-only, with the keyword parameters as they stand.  This is how the lists are pinned against the reference's
 * own BiddingOutcomes (adcraft/bidding_simulation.py:10-38,124-147; tests/golden/g3_*.json, g8_env_episodes.json). */
int adc_engine_outcomes_replay_tape(adc_engine *e, int32_t env, const float *bids_k, float budget, const adc_tape *tape, int64_t capacity,
                                    int32_t *keyword, int32_t *timestep, double *cost, double *revenue, int64_t *count, int32_t *share_volume_k);

/* ---- standalone auction clearing (adcraft/synthetic_kw_helpers.py:116-180) ------------------------ */
/* other_bids: host double [n_auctions][n_bidders]; placements/costs: host, capacity n_auctions.
 * Returns the impression count in *impressions.  num_winners + n must be <= 32. */
int adc_nth_price_auction(int device_id, double bid, const double *other_bids, int32_t n_auctions, int32_t n_bidders,
                          int32_t n, int32_t num_winners, int32_t *impressions, int32_t *placements, double *costs);

/* ---- scalar entry points mirroring `adcraft.rust` (src/lib.rs, function by function) -------------- */
double adc_sigmoid(double x, double s, double t);                               /* src/lib.rs:79-83,290-294 */
double adc_clamp(double x, double lo, double hi);                               /* probify_float, :86-90 */
double adc_threshold_sigmoid(double p, double impression_thresh,
                             double impression_bid_intercept, double impression_slope);   /* :93-105 */
double adc_sum_f64(const double *x, int64_t n);                                 /* sum_array / sum_list, :108-116 */
int64_t adc_count_true(const uint8_t *x, int64_t n);                            /* sum_array_bool / sum_list_bool */
/* samplers: the reference draws from an unseeded thread_rng (src/lib.rs:25,61,75,320); these draw from a
 * Philox stream keyed by (seed, counter) so callers can be reproducible. */
uint64_t adc_nonneg_int_normal(double mean, double std, uint64_t seed, uint64_t counter);   /* :314-325 */
uint64_t adc_binomial(uint64_t n, double p, uint64_t seed, uint64_t counter);               /* :70-76 */
int adc_cost_create(double x, int64_t n, uint64_t seed, uint64_t counter, double *out_n);   /* :54-67 */
/* diagnostic: the word-space thresholds k_step_implicit_fast resolves a keyword's auctions with (adc_law.h
 * win_intervals): word w is a clicked win iff (w - out4[0]) < out4[1], an unclicked win iff (w - out4[2]) < out4[3]
 * (uint32 arithmetic) - equivalent, word for word, to sampling the competitor's bid and running the reference's
 * nth_price_auction env path on it (adcraft/synthetic_kw_helpers.py:116-180: win iff bid > competitor) */
int adc_auction_word_intervals(float bid, float cost_loc, float cost_scale, float buyside_ctr, uint32_t *out4);
/* diagnostic: the conservative BRACKETS of those two intervals that k_step_implicit_sparse classifies a sparse keyword's
 * auctions with (adc_law.h win_brackets): out8 = {outer c_lo, c_w, n_lo, n_w, inner c_lo, c_w, n_lo, n_w}; a word inside an
 * inner interval wins, a word outside the outer ones loses, the rest is resolved from the sampled competitor bid.
 * adc_check_win_brackets verifies inner <= exact <= outer for n keywords (brackets8 NULL: the host's own evaluation;
 * otherwise e.g. adc_debug_win_brackets_device's) and returns the number of violations (0 expected). */
int adc_auction_word_brackets(float bid, float cost_loc, float cost_scale, float buyside_ctr, uint32_t *out8);
int64_t adc_check_win_brackets(int64_t n, const float *bid, const float *cost_loc, const float *cost_scale, const float *buyside_ctr,
                               const uint32_t *brackets8, int64_t *first_bad, double *ambiguous_words);
/* one keyword of the default constructor's keyword set, on the host: exactly what adc_engine_generate_explicit_keywords writes for
 * keyword `keyword` of an env whose Philox key is `key` (adc_engine_get_rng_state) - sample_random_keywords' law
 * (gymnasium_kw_utils.py:113-156); out8 in adc_param order */
int adc_sample_random_keyword(uint64_t key, uint32_t keyword, uint32_t serial, float *out8);
/* diagnostic: the stream's generator (Philox4x32, the stream's round count) evaluated on the device for n counters ctr4[n][4]
 * and keys key2[n][2] -> out4[n][4]; tests compare it with the CPU battery's generator (oracle/stream_battery.c) */
int adc_debug_philox_device(int device_id, int64_t n, const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4);
/* diagnostic: the reference's float64 budget chain `remaining -= sum(costs)` (bidding_simulation.py:225) over n cell sums x, as the
 * budget-exact kernels evaluate it - 64 rounded subtractions at a time by an integer prefix scan inside the running value's binade
 * (parts/common.inc chain_subtract_wave) -> out2[0]; and by the plain chain of n rounded subtractions on the device -> out2[1].
 * The two are the same float64, bit for bit, for every input. */
int adc_debug_chain_device(int device_id, double r0, int64_t n, const double *x, double *out2);
/* counters of k_step_click_walk on the engine's device since the library was loaded (or the last call with reset != 0):
 * stats[0] env-steps walked, [1] handed to the row kernel because the click list overflowed, [2] because the campaign
 * stopped, [3] for another reason (budget <= 0, a keyword-day above 2^22 cents in metric mode).  Test / measurement aid. */
int adc_debug_walk_stats(adc_engine *e, int64_t stats[4], int reset);
/* env-days k_tail_or_flag handed to k_step_rest_of_day at once, without the row kernel (a budget that ran out within the first
 * cells of the previous day), by THIS engine since it was created (or the last call with reset != 0).  Test / measurement aid. */
int adc_debug_direct_days(adc_engine *e, int64_t *env_days, int reset);
int adc_debug_win_brackets_device(int device_id, int64_t n, const float *bid, const float *cost_loc, const float *cost_scale,
                                  const float *buyside_ctr, uint32_t *out8);

#ifdef __cplusplus
}
#endif
#endif /* ADCRAFT_ENGINE_H */
