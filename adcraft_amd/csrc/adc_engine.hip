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
// One translation unit (so the device log table and all inlining stay local without -fgpu-rdc), in parts:
//   parts/common.inc              device view of the engine (SoA pointers), wave scan/sum helpers, log table, drift
//   parts/kernel_fast.inc         k_step_implicit_fast - the budget-free pass for dense keyword sets.  One workgroup per (env,
//        256-keyword tile): phase 1 one lane per keyword (coalesced SoA loads, pending drift, volume draw, the keyword's two
//        exact win intervals in word space and its law records in LDS); phase 2 work items of 16 / 8 / 4 whole Philox calls
//        dealt lane-major, one word per auction, an auction = two subtract-and-compare pairs (stage A), clicked wins
//        compacted through per-wave LDS rings for the price + conversion / revenue call on full wavefronts (stage B),
//        integer-cent totals by LDS atomics; phase 3 coalesced observation stores.  Ignores the budget; exact whenever
//        the day's spend stays below it.
//   parts/kernel_sparse.inc       k_step_implicit_sparse - the same pass for keyword sets with few auctions per keyword (the
//        host's volume hint): auctions classified against conservative win BRACKETS (float estimates with a guaranteed
//        slack; the ~2e-5 of words in between take the long way), work items located through a byte list in LDS,
//        parameters / env header / metric sums software-pipelined across the tiles of a workgroup.
//   parts/kernel_exact_rows.inc   k_step_exact_rows - one workgroup per env: envs whose fast-pass spend reached the
//        budget are re-run in the reference's order, a sub-timestep row of K cells at a time (parallel cell
//        statistics, budget walk by prefix scans, ring-compacted conversions); then the step tail.
//        Since round 3 it stops at the row after the one the budget bound in and parks the env for k_step_rest_of_day (same file):
//        the remaining rows' auctions in whole Philox calls, wins counted by exact word intervals, the few cells that may hold
//        an affordable click walked in order.
//   parts/kernel_click_walk.inc   k_step_click_walk - binding budgets, K <= 256: the fast pass lists its clicked wins for envs whose
//        budget bound the day before; this kernel sorts the list into the reference's order (counting sort over the 24 x K
//        cells in LDS) and walks it - paid in full while the running total fits, click by click in the reference's float
//        arithmetic after - instead of re-running the day row by row.  Hands anything unusual to k_step_exact_rows.
//   parts/kernel_exact_serial.inc step_tail + k_step_exact - one wavefront per env walking cells serially: TAPE
//        replay of the reference's recorded variates, the EXPLICIT / IMPLICIT_GENERAL hand-overs, IMPLICIT with K > 2048, and the
//        read-only replay that regenerates a step's per-click lists (adc_engine_outcomes_replay).
//   parts/kernel_explicit_fast.inc the other two keyword models, keyword-parallel: k_step_explicit_fast, k_step_general_fast (the
//        reference's default ImplicitKeyword: top bids drawn as order statistics), k_step_float_day, k_step_explicit_rows.
//   parts/kernels_misc.inc        drift, metric sums, ideal profit, keyword generation, reset, synthetic actions,
//        flat observations/actions, nth_price_auction.
//   parts/kernels_policy.inc      the callers of the step on the device: the zero-margin agent, the per-step ideal profit on
//        the curves' contender lists (k_curve_contenders, k_ideal_from_contenders), the oracle bidder, per-env AKNCP / NCP.
//   parts/host_api.inc            the engine object and the extern "C" entry points.
//
// No CPU path exists in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <type_traits>
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
#include "parts/common.inc"
#include "parts/kernel_fast.inc"
#include "parts/kernel_sparse.inc"
#include "parts/kernel_exact_rows.inc"
#include "parts/kernel_click_walk.inc"
#include "parts/kernel_exact_serial.inc"
#include "parts/kernel_explicit_fast.inc"
#include "parts/kernels_misc.inc"
#include "parts/kernels_policy.inc"
}  // namespace adck
using namespace adck;

#include "parts/host_api.inc"
#include "parts/comm_api.inc"
