// throughput.cpp -- the headline metric (M paths/s of the ZBC option kernel, reference README.md:44: "1.97 ms, 1064 M
// paths/sec" for one launch of 2^20 pairs) measured by a plain C++ program over the C ABI: no Python, no torch.
// A batch of independent pricing calls (the reference's 20-run validation seeds, src/2_option_pricing.cu:229) goes
//   (a) through hw_zbc_moments one synchronous call after another   -- what the reference's loop does, and
//   (b) through hw_pipe_zbc_moments with two launches in flight     -- one copy, one synchronisation per batch,
// and both must produce the same sums bit for bit.
//   bin/throughput [--paths-log2 20] [--steps 1000] [--calls 20] [--queues 2] [--batches 7] [--precision f32|f64]
//                  [--theta-profile head|readme] [--device 0]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/hw_mc.h"

static void die(const char* what)
{
    printf("HIP Error [%s]: %s\n", what, hw_last_error());
    exit(1);
}

int main(int argc, char** argv)
{
    int paths_log2 = 20, steps = 1000, calls = 20, queues = 2, batches = 7, device = 0, precision = HW_F32;
    const char* profile = "head";
    for (int i = 1; i < argc; i++) {
        auto val = [&](const char* flag) -> const char* {
            if (strcmp(argv[i], flag) || i + 1 >= argc) return nullptr;
            return argv[++i];
        };
        if (const char* v = val("--paths-log2")) paths_log2 = atoi(v);
        else if (const char* v2 = val("--steps")) steps = atoi(v2);
        else if (const char* v3 = val("--calls")) calls = atoi(v3);
        else if (const char* v4 = val("--queues")) queues = atoi(v4);
        else if (const char* v5 = val("--batches")) batches = atoi(v5);
        else if (const char* v6 = val("--device")) device = atoi(v6);
        else if (const char* v7 = val("--theta-profile")) profile = v7;
        else if (const char* v8 = val("--precision")) precision = !strcmp(v8, "f64") ? HW_F64 : HW_F32;
        else { printf("unknown or incomplete flag %s\n", argv[i]); return 2; }
    }
    if (paths_log2 < 0 || paths_log2 > 30 || calls < 1 || calls > 4096 || batches < 1) { printf("bad arguments\n"); return 2; }
    const int64_t n = (int64_t)1 << paths_log2;

    hw_params p;
    if (hw_default_params(&p, profile)) die("hw_default_params");
    p.n_steps = steps;
    p.precision = precision;
    std::vector<float> P(p.n_mat), f(p.n_mat);
    if (hw_analytic_curve(&p, P.data(), f.data())) die("hw_analytic_curve");     // synthetic market input: the exact curve

    hw_ctx* ctx = nullptr;
    if (hw_create(&p, device, &ctx)) die("hw_create");
    if (hw_set_market(ctx, P.data(), f.data(), p.n_mat)) die("hw_set_market");
    hw_pipe* pipe = nullptr;
    if (hw_pipe_create(&p, device, queues, &pipe)) die("hw_pipe_create");
    if (hw_pipe_set_market(pipe, P.data(), f.data(), p.n_mat)) die("hw_pipe_set_market");
    hw_device_info info;
    hw_get_device_info(ctx, &info);

    std::vector<hw_zbc_request> rq(calls);
    for (int k = 0; k < calls; k++) {
        memset(&rq[k], 0, sizeof rq[k]);
        rq[k].n_pairs = n;
        rq[k].seed = 1234ull + 12345ull * (uint64_t)k;                           // 2_option:229
        rq[k].S1 = 5.0; rq[k].S2 = 10.0; rq[k].K = std::exp(-0.1);
        rq[k].drift_mode = HW_DRIFT_BASE; rq[k].stepper = HW_STEPPER_SUPERPOSED;
    }
    float tuned = 0.0f;
    if (hw_pipe_tune(pipe, &rq[0], 0, &tuned)) die("hw_pipe_tune");              // untimed: clocks up, stream pair chosen

    std::vector<double> serial_sums((size_t)calls * 5), piped_sums((size_t)calls * 5);
    std::vector<double> t_serial, t_piped;
    for (int b = 0; b < batches; b++) {
        float wall = 0.0f;
        if (hw_pipe_zbc_moments(pipe, rq.data(), calls, piped_sums.data(), &wall)) die("hw_pipe_zbc_moments");
        t_piped.push_back(wall / calls);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < calls; k++)
            if (hw_zbc_moments(ctx, &rq[k], &serial_sums[(size_t)5 * k], nullptr)) die("hw_zbc_moments");
        t_serial.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / calls);
    }
    if (memcmp(serial_sums.data(), piped_sums.data(), serial_sums.size() * sizeof(double)) != 0) {
        printf("FAILED: the pipe's sums differ from the synchronous calls'\n");
        return 1;
    }
    auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double ms_s = median(t_serial), ms_p = median(t_piped);
    double res[9];
    hw_cv_price(&piped_sums[0], 2 * n, P[p.n_mat - 1], res);
    printf("device: %s\n", info.name);
    printf("workload: %d pricing calls of 2^%d antithetic pairs x %d-step grid (%d executed), %s, theta profile %s\n", calls, paths_log2,
           steps, steps / 2, precision == HW_F64 ? "f64" : "f32", profile);
    printf("synchronous calls (hw_zbc_moments):        %.4f ms per call   %10.1f M paths/s\n", ms_s, 2.0 * n / ms_s / 1e3);
    printf("%d queues in flight (hw_pipe_zbc_moments):  %.4f ms per call   %10.1f M paths/s   (%.2fx the reference's 1064 M on a V100)\n",
           queues, ms_p, 2.0 * n / ms_p / 1e3, 2.0 * n / ms_p / 1e3 / 1064.0);
    printf("sums identical: yes;  ZBC price of call 0: %.8f  beta* %.6f\n", res[6], res[4]);
    printf("{\"metric\": \"M paths/sec (ZBC option, %d steps)\", \"pipelined\": %.2f, \"synchronous\": %.2f, \"ms_per_call_pipelined\": %.5f, "
           "\"ms_per_call_synchronous\": %.5f, \"calls\": %d, \"queues\": %d, \"pairs_per_call\": %lld}\n",
           steps, 2.0 * n / ms_p / 1e3, 2.0 * n / ms_s / 1e3, ms_p, ms_s, calls, queues, (long long)n);
    hw_pipe_destroy(pipe);
    hw_destroy(ctx);
    return 0;
}
