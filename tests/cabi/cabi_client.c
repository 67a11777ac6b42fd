/* A plain-C client of include/dhmc.h: what a cgo / ccall / JNI binding sees.  Runs 8 chains of a 100-dim standard
 * normal through warmup-style calls and prints a checksum of the draws; tests/test_gpu_cabi.py compares it with the
 * ctypes path.  Build: gcc -std=c99 cabi_client.c -I../../include -L../../dynamichmc.jl_amd/lib -ldhmc_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dhmc.h"

int main(void) {
    dhmc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.dim = 100; cfg.chains = 8; cfg.chain_offset = 0;
    cfg.metric = DHMC_METRIC_DIAG; cfg.target = DHMC_TARGET_STD_NORMAL;
    cfg.max_depth = 10; cfg.min_delta = -1000.0; cfg.seed = 0x23EF614Dull;
    dhmc_ctx* ctx = NULL;
    int rc = dhmc_create(&cfg, &ctx);
    if (rc != DHMC_OK) { printf("dhmc_create failed: %d\n", rc); return 2; }
    if ((rc = dhmc_init(ctx, NULL, 0)) != DHMC_OK) { printf("init: %d %s\n", rc, dhmc_last_error(ctx)); return 2; }
    if ((rc = dhmc_find_initial_stepsize(ctx, NULL)) != DHMC_OK) { printf("search: %d\n", rc); return 2; }
    dhmc_dual_averaging da;
    memset(&da, 0, sizeof da);
    da.delta = 0.8; da.gamma = 0.05; da.kappa = 0.75; da.t0 = 10; da.init = 1; da.finalize = 1;
    const int64_t n = 30;
    double* draws = (double*)malloc(sizeof(double) * 8 * n * 100);
    int64_t* steps = (int64_t*)malloc(sizeof(int64_t) * 8 * n);
    dhmc_outputs out;
    memset(&out, 0, sizeof out);
    out.on_device = 0; out.draws = draws; out.steps = steps;
    if ((rc = dhmc_run(ctx, n, &da, &out)) != DHMC_OK) { printf("run(adapt): %d\n", rc); return 2; }
    if ((rc = dhmc_update_metric_diag(ctx, draws, n, 0.0, 0)) != DHMC_OK) { printf("metric: %d\n", rc); return 2; }
    if ((rc = dhmc_run(ctx, n, NULL, &out)) != DHMC_OK) { printf("run: %d\n", rc); return 2; }
    double sum = 0.0; long long nsteps = 0;
    for (int64_t i = 0; i < 8 * n * 100; ++i) sum += draws[i];
    for (int64_t i = 0; i < 8 * n; ++i) nsteps += steps[i];
    double eps[8];
    dhmc_get_stepsize(ctx, eps, 0);
    printf("version %s\nchecksum %.17g\nsteps %lld\neps0 %.17g\nleapfrogs %llu\n", dhmc_version(), sum, nsteps, eps[0],
           (unsigned long long)dhmc_last_run_leapfrogs(ctx));
    /* a tuning stage whose metric comes from a window the kernels accumulate: no outputs at all during the stage */
    if (dhmc_metric_window_count(ctx) != -1) { printf("window open before begin\n"); return 2; }
    if ((rc = dhmc_metric_window_begin(ctx)) != DHMC_OK) { printf("window begin: %d\n", rc); return 2; }
    if ((rc = dhmc_run(ctx, n, &da, NULL)) != DHMC_OK) { printf("run(window): %d\n", rc); return 2; }
    if (dhmc_metric_window_count(ctx) != n) { printf("window count\n"); return 2; }
    if ((rc = dhmc_update_metric_diag_window(ctx, 0.0)) != DHMC_OK) { printf("window update: %d\n", rc); return 2; }
    if (dhmc_update_metric_diag_window(ctx, 0.0) != DHMC_ERR_INVALID_ARGUMENT) { printf("closed window accepted\n"); return 2; }
    if ((rc = dhmc_run(ctx, n, NULL, &out)) != DHMC_OK) { printf("run after window: %d\n", rc); return 2; }
    sum = 0.0;
    for (int64_t i = 0; i < 8 * n * 100; ++i) sum += draws[i];
    printf("checksum_after_window %.17g\n", sum);
    free(draws); free(steps);
    return dhmc_destroy(ctx) == DHMC_OK ? 0 : 2;
}
