// Compile-only check (tests/test_host_cpp.py) that the call shapes INTEGRATION.md shows a maintainer of the reference match include/hw_mc.h.
#include "hw_mc.h"
#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
int main() {
    const int N_MAT = 101; const long long N_PATHS = 1 << 20; float h_P[101], h_f[101], h_show_paths[32 * 1001];
    hw_params p;  hw_default_params(&p, "head");
    hw_ctx* ctx;  hw_create(&p, -1, &ctx);
    hw_set_market(ctx, h_P, h_f, N_MAT);
    hw_zbc_request rq = {0};
    rq.n_pairs = N_PATHS;  rq.seed = time(NULL) + 54321;
    rq.S1 = 5.0; rq.S2 = 10.0; rq.K = expf(-0.1f);
    double sums[5]; float sim_ms;
    if (hw_zbc_moments(ctx, &rq, sums, &sim_ms) != HW_OK) { printf("HIP Error: %s\n", hw_last_error()); exit(1); }
    double res[9]; hw_cv_price(sums, 2LL * N_PATHS, h_P[N_MAT-1], res);
    double sums20[20 * 5]; float ms; hw_zbc_moments_runs(ctx, &rq, 20, 12345, sums20, &ms);
    double P_sum[101]; unsigned long long seed = 1;
    hw_zcb_curve(ctx, N_PATHS, time(NULL), 0, 0.0, P_sum, &sim_ms);
    hw_average_and_forward(&p, P_sum, 2LL * N_PATHS, h_P, h_f);
    hw_sample_paths(ctx, 32, seed, 0, h_show_paths);
    double K = rq.K, s; hw_vega_pathwise(ctx, N_PATHS, seed, 0, 5.0, 10.0, K, 0, &s, &ms);
    const double H_SIGMA = 0.1;
    double sig[2] = {H_SIGMA - 1e-3, H_SIGMA + 1e-3};  int32_t dm[2] = {HW_DRIFT_SHIFTED, HW_DRIFT_SHIFTED};
    double sc[10]; hw_zbc_scenarios(ctx, &rq, 2, sig, dm, sc, NULL);
    unsigned long long base = 5; double v20[20]; hw_vega_pathwise_runs(ctx, N_PATHS, base, 0, 5.0, 10.0, K, 0, 20, 982451653ull, v20, &ms);
    hw_ctx* ctxs[1] = {ctx}; hw_comm* comm; hw_comm_create(1, ctxs, &comm); float wall_ms;
    hw_comm_zbc_moments(comm, &rq, sums, &wall_ms);
    hw_comm_zbc_moments_runs(comm, &rq, 20, 12345ull, sums20, &wall_ms);          // the 20-run validation over the fleet
    hw_comm_destroy(comm);
    hw_comm_create_ex(1, ctxs, HW_COMM_HOST_SUM, &comm); hw_comm_destroy(comm);   // without RCCL
    hw_pipe* pipe;  hw_pipe_create(&p, -1, /*n_queues*/ 2, &pipe);
    hw_pipe_set_market(pipe, h_P, h_f, N_MAT);
    hw_zbc_request rqs[20];
    for (int k = 0; k < 20; k++) { rqs[k] = rq; rqs[k].seed = rq.seed + 12345ull * k; }
    hw_pipe_tune(pipe, &rqs[0], 0, NULL);
    hw_pipe_zbc_moments(pipe, rqs, 20, sums20, &wall_ms);
    hw_pipe_destroy(pipe);
    hw_destroy(ctx);
    return 0;
}
