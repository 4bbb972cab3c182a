/* Plain C99 consumer of include/adcraft_engine.h: creates an engine, uploads one keyword set, runs a few
 * BiddingSimulation steps for 4 environments x 8 keywords and prints the observations.
 *   gcc -std=c99 -Iinclude examples/c_abi_minimal.c -Ladcraft_amd/lib -ladcraft_hip -Wl,-rpath,$PWD/adcraft_amd/lib -o c_abi_minimal
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "adcraft_engine.h"

#define N 4
#define K 8
#define CHECK(x) do { int rc_ = (x); if (rc_ != ADC_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, adc_last_error()); return 1; } } while (0)

int main(void)
{
    adc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.num_envs = N;
    cfg.num_keywords = K;
    cfg.model = ADC_MODEL_IMPLICIT;
    cfg.max_days = 3;
    cfg.loss_threshold = 10000.0;
    cfg.impression_thresh = 0.05f;
    cfg.seed = 7;
    adc_engine *e = NULL;
    CHECK(adc_engine_create(&cfg, &e));

    /* keyword state: volume N(64, 8), competitor bid |Laplace(0.55, 0.08)|, ctr 0.5, cvr 0.8, revenue N(1.0, 0.15) */
    const float vals[ADC_P_COUNT] = {64.f, 8.f, 0.55f, 0.08f, 0.5f, 0.8f, 1.0f, 0.15f};
    float plane[N * K];
    for (int p = 0; p < ADC_P_COUNT; ++p) {
        for (int i = 0; i < N * K; ++i) plane[i] = vals[p];
        CHECK(adc_engine_set_params(e, p, plane));
    }
    CHECK(adc_engine_reset(e, NULL, NULL));

    float bids[N * K], budget[N];
    int32_t imp[N * K], clk[N * K], conv[N * K], days[N];
    float cost[N * K], rev[N * K];
    double reward[N], cum[N];
    uint8_t term[N], trunc[N];
    adc_step_out out = {imp, clk, conv, cost, rev, reward, cum, days, term, trunc};
    for (int i = 0; i < N * K; ++i) bids[i] = 0.40f + 0.05f * (float)(i % K);
    for (int i = 0; i < N; ++i) budget[i] = i == 0 ? 5.0f : 1000.0f;      /* env 0 hits its budget */

    for (int step = 0; step < 3; ++step) {
        CHECK(adc_engine_step(e, bids, budget, &out));
        for (int env = 0; env < N; ++env) {
            long ti = 0, tc = 0;
            double spent = 0;
            for (int k = 0; k < K; ++k) { ti += imp[env * K + k]; tc += clk[env * K + k]; spent += cost[env * K + k]; }
            printf("step %d env %d: impressions %ld clicks %ld spent %.2f reward %.2f cum %.2f day %d%s%s\n", step, env, ti, tc,
                   spent, reward[env], cum[env], days[env], term[env] ? " terminated" : "", trunc[env] ? " truncated" : "");
            if (env == 0 && spent > 5.0 + 1e-6) { fprintf(stderr, "budget exceeded\n"); return 2; }
        }
    }
    adc_engine_destroy(e);
    printf("ok\n");
    return 0;
}
