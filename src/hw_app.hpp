// hw_app.hpp -- shared support for the three drop-in executables (bin/q1, bin/q2, bin/q3).
// They keep the reference's process surface -- same make targets, same files under data/, same
// y/n prompts on stdin when no flag is given -- and call the MI355X engine only through the C ABI
// (include/hw_mc.h).  File formats follow include/output.cuh of the reference (json header
// output.cuh:8-33, arrays :41-50, performance :53-60, csv :62-96, summary :128-167) and the raw
// float32 hand-off of include/common.cuh:144-169.
//
// Flags (all optional; the reference configures everything at compile time, common.cuh:16-39):
//   --paths-log2 K   antithetic pairs / paths per run = 2^K        (default 20 = N_PATHS)
//   --steps N        grid steps over T_FINAL = 10                  (default 1000 = N_STEPS)
//   --seed S         base seed (default: time(NULL)-derived, as the reference)
//   --theta-profile head|readme     theta intercept after t = 5: 0.019 (source) | 0.014 (README)
//   --precision f32|f64             per-path stepper arithmetic
//   --stepper superposed|reference  ZBC kernel arithmetic
//   --market mc|exact               q2/q3: read q1's data/P.bin,f.bin (default) | exact curve
//   --gpus G         shard the paths over G GPUs (one engine context per device, sums added on host)
//   --device D       first device index
//   --yes / --no     answer every y/n prompt without reading stdin
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include "../include/hw_mc.h"

namespace app {

struct Options {
    int paths_log2 = 20;
    int steps = 1000;
    bool have_seed = false;
    uint64_t seed = 0;
    std::string profile = "head";
    int precision = HW_F32;
    int stepper = HW_STEPPER_SUPERPOSED;
    std::string market = "mc";
    int gpus = 1, device = 0;
    int spinup_ms = 150;   // untimed repeats of the first launch before the timed one: the chip's clock ramps over ~100 ms of load
    int reduce = -1;   // cross-device sum: -1 automatic (RCCL when --gpus > 1), 0 host threads, 1 RCCL all-reduce
    int answer = -1;   // -1 ask on stdin, 0 no, 1 yes
    int64_t n() const { return (int64_t)1 << paths_log2; }
};

[[noreturn]] inline void die(const char* what, const char* text = nullptr)
{
    // check_cuda, common.cuh:114-120.  hw_last_error() is per thread: a caller that ran the engine on a worker
    // thread passes the text it captured there.
    printf("HIP Error [%s]: %s\n", what, text ? text : hw_last_error());
    exit(1);
}
#define HW_CHECK(call) do { if ((call) != HW_OK) app::die(#call); } while (0)

inline Options parse(int argc, char** argv)
{
    Options o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); }
            return argv[++i];
        };
        if (a == "--paths-log2") o.paths_log2 = atoi(val());
        else if (a == "--steps") o.steps = atoi(val());
        else if (a == "--seed") { o.seed = strtoull(val(), nullptr, 0); o.have_seed = true; }
        else if (a == "--theta-profile") o.profile = val();
        else if (a == "--precision") o.precision = !strcmp(val(), "f64") ? HW_F64 : HW_F32;
        else if (a == "--stepper") o.stepper = !strcmp(val(), "reference") ? HW_STEPPER_REFERENCE : HW_STEPPER_SUPERPOSED;
        else if (a == "--market") o.market = val();
        else if (a == "--gpus") o.gpus = atoi(val());
        else if (a == "--device") o.device = atoi(val());
        else if (a == "--reduce") o.reduce = !strcmp(val(), "rccl") ? 1 : 0;
        else if (a == "--spinup-ms") o.spinup_ms = atoi(val());
        else if (a == "--yes") o.answer = 1;
        else if (a == "--no") o.answer = 0;
        else { fprintf(stderr, "unknown flag %s (see src/hw_app.hpp)\n", a.c_str()); exit(2); }
    }
    if (o.paths_log2 < 0 || o.paths_log2 > 40 || o.gpus < 1 || o.gpus > 64) { fprintf(stderr, "bad --paths-log2/--gpus\n"); exit(2); }
    return o;
}

inline hw_params make_params(const Options& o)
{
    hw_params p;
    if (hw_default_params(&p, o.profile.c_str()) != HW_OK) die("hw_default_params");
    p.n_steps = o.steps;
    p.precision = o.precision;
    return p;
}

// y/n prompt of 2_option:495-500 / 3_sens:725-730: reads one non-blank character from stdin.
inline bool ask(const Options& o, const char* prompt)
{
    printf("%s", prompt);
    fflush(stdout);
    if (o.answer >= 0) { printf("%c\n", o.answer ? 'y' : 'n'); return o.answer == 1; }
    char c = 'n';
    if (scanf(" %c", &c) != 1) return false;
    return c == 'y' || c == 'Y';
}

inline void ensure_dirs() { mkdir("data", 0755); mkdir("plots", 0755); }

// ---- raw float32 arrays: save_array / load_array, common.cuh:144-169 ------------------------
inline void save_array(const char* path, const float* data, int n)
{
    FILE* f = fopen(path, "wb");
    if (!f) { printf("Error: Cannot open %s for writing\n", path); exit(1); }
    fwrite(data, sizeof(float), n, f);
    fclose(f);
    printf("Saved %s (%d floats)\n", path, n);
}
inline void load_array(const char* path, float* data, int n)
{
    FILE* f = fopen(path, "rb");
    if (!f) { printf("Error: Cannot open %s\n", path); printf("Did you run Q1 first?\n"); exit(1); }
    const size_t got = fread(data, sizeof(float), n, f);
    fclose(f);
    if ((int)got != n) { printf("Error: Expected %d floats, got %zu\n", n, got); exit(1); }
    printf("Loaded %s (%d floats)\n", path, n);
}

// ---- JSON / CSV / summary writers (output.cuh) -----------------------------------------------
struct Json {
    FILE* f = nullptr;
    bool open(const char* path, const char* task, const hw_params& p, int64_t n_paths)
    {
        f = fopen(path, "w");
        if (!f) { printf("Error: Cannot create %s\n", path); return false; }
        time_t now = time(nullptr);
        std::string ts = ctime(&now);
        if (!ts.empty() && ts.back() == '\n') ts.pop_back();
        fprintf(f, "{\n  \"task\": \"%s\",\n  \"timestamp\": \"%s\",\n  \"parameters\": {\n", task, ts.c_str());
        fprintf(f, "    \"N_PATHS\": %lld,\n    \"N_STEPS\": %d,\n    \"N_MAT\": %d,\n", (long long)n_paths, p.n_steps, p.n_mat);
        fprintf(f, "    \"T_FINAL\": %.1f,\n    \"a\": %.2f,\n    \"sigma\": %.2f,\n    \"r0\": %.4f\n  },\n", p.T_final, p.a, p.sigma, p.r0);
        return true;
    }
    void array(const char* name, const float* d, int n)
    {
        fprintf(f, "  \"%s\": [", name);
        for (int i = 0; i < n; i++) {
            if (i % 10 == 0) fprintf(f, "\n    ");
            fprintf(f, "%.8f%s", d[i], i < n - 1 ? ", " : "");
        }
        fprintf(f, "\n  ]");
    }
    void performance(float ms, double n_paths)
    {
        fprintf(f, "  \"performance\": {\n    \"simulation_time_ms\": %.2f,\n    \"throughput_Mpaths_per_sec\": %.2f\n  }",
                ms, (n_paths / ms) / 1000.0);
    }
    void raw(const char* s) { fputs(s, f); }
    void close() { fprintf(f, "}\n"); fclose(f); f = nullptr; }
};

inline void csv_timeseries(const char* path, const char* header, const float* d, int n, float spacing)
{
    FILE* f = fopen(path, "w");
    if (!f) { printf("Error: Cannot create %s\n", path); return; }
    fprintf(f, "T,%s\n", header);
    for (int i = 0; i < n; i++) fprintf(f, "%.4f,%.8f\n", i * spacing, d[i]);
    fclose(f);
    printf("Saved %s\n", path);
}
inline void csv_comparison(const char* path, const float* x, const float* y1, const float* y2, const char* xn,
                           const char* y1n, const char* y2n, int n)
{
    FILE* f = fopen(path, "w");
    if (!f) { printf("Error: Cannot create %s\n", path); return; }
    fprintf(f, "%s,%s,%s\n", xn, y1n, y2n);
    for (int i = 0; i < n; i++) fprintf(f, "%.4f,%.8f,%.8f\n", x[i], y1[i], y2[i]);
    fclose(f);
    printf("Saved %s\n", path);
}

static const char* kRule = "================================================================================\n";
inline void summary_init(const char* path, const hw_params& p, int64_t n_paths)
{
    FILE* f = fopen(path, "w");
    if (!f) { printf("Error: Cannot create %s\n", path); return; }
    time_t now = time(nullptr);
    fprintf(f, "%sHULL-WHITE MODEL SIMULATION RESULTS\n%sGenerated: %s\nParameters:\n", kRule, kRule, ctime(&now));
    fprintf(f, "  N_PATHS = %lld (x2 antithetic = %lld effective)\n  N_STEPS = %d\n  N_MAT = %d\n", (long long)n_paths,
            (long long)(2 * n_paths), p.n_steps, p.n_mat);
    fprintf(f, "  T_FINAL = %.1f years\n  a = %.2f, sigma = %.2f, r0 = %.4f\n", p.T_final, p.a, p.sigma, p.r0);
    fclose(f);
    printf("Initialized %s\n", path);
}
inline FILE* summary_section(const char* path, const char* title)
{
    FILE* f = fopen(path, "a");
    if (!f) { printf("Error: Cannot open %s\n", path); return nullptr; }
    fprintf(f, "\n%s%s\n%s", kRule, title, kRule);
    return f;
}

// ---- engines: one context per GPU; paths sharded by global id, sums added on the host --------
struct Fleet {
    std::vector<hw_ctx*> ctx;
    hw_comm* comm = nullptr;   // non-null: shard + ONE grouped ncclAllReduce inside the library (hw_comm_*)
    hw_params p{};
    void create(const hw_params& params, int gpus, int first_device, int reduce = -1)
    {
        p = params;
        for (int g = 0; g < gpus; g++) {
            hw_ctx* c = nullptr;
            if (hw_create(&p, gpus == 1 && first_device < 0 ? -1 : first_device + g, &c) != HW_OK) die("hw_create");
            ctx.push_back(c);
        }
        hw_device_info di;
        HW_CHECK(hw_get_device_info(ctx[0], &di));
        printf("Using GPU %d: %s, %d CUs, %.2f GB%s\n\n", di.device, di.name, di.compute_units, di.hbm_bytes / 1e9,
               gpus > 1 ? " (+ more devices)" : "");
        if (reduce == 1 || (reduce < 0 && gpus > 1)) {
            HW_CHECK(hw_comm_create((int32_t)ctx.size(), ctx.data(), &comm));
            printf("Cross-device sums: RCCL all-reduce over %d device%s\n\n", gpus, gpus > 1 ? "s" : "");
        }
    }
    void set_market(const float* P, const float* f) { for (hw_ctx* c : ctx) HW_CHECK(hw_set_market(c, P, f, p.n_mat)); }
    void destroy() { hw_comm_destroy(comm); comm = nullptr; for (hw_ctx* c : ctx) hw_destroy(c); ctx.clear(); }
    // Warm-up: at least one call of fn, repeated until `ms` of wall time have passed.  The reference times a single
    // cold launch (2_option:131-144); a cold MI355X runs the first ~100 ms of work 10-15 % below its sustained clock.
    template <typename F>
    void warm(int ms, F fn)
    {
        const auto t0 = std::chrono::steady_clock::now();
        do { fn(); } while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < ms);
    }
    // shard [0, n) over the devices; returns {offset, count} of device g
    void shard(int64_t n, int g, int64_t* off, int64_t* cnt) const
    {
        const int64_t G = (int64_t)ctx.size(), base = n / G, rem = n % G;
        *off = g * base + std::min<int64_t>(g, rem);
        *cnt = base + (g < rem ? 1 : 0);
    }
    // Run fn(g, offset, count) for every device concurrently (one host thread per context, as the
    // C ABI requires); with one device it runs inline.
    template <typename F>
    void for_each_device(int64_t n, F fn)
    {
        if (ctx.size() == 1) { fn(0, (int64_t)0, n); return; }
        std::vector<std::thread> th;
        for (size_t g = 0; g < ctx.size(); g++) {
            int64_t off, cnt;
            shard(n, (int)g, &off, &cnt);
            th.emplace_back([=, &fn]() { fn((int)g, off, cnt); });
        }
        for (auto& t : th) t.join();
    }
    // ZBC moments over n pairs (all devices), kernel time = slowest device
    void zbc(hw_zbc_request r, double out[5], float* ms)
    {
        if (comm) { HW_CHECK(hw_comm_zbc_moments(comm, &r, out, ms)); return; }
        const size_t G = ctx.size();
        std::vector<double> part(5 * G, 0.0);
        std::vector<float> t(G, 0.0f);
        std::vector<int> rc(G, 0);
        std::vector<std::string> err(G);
        const uint64_t base_off = r.path_offset;
        for_each_device(r.n_pairs, [&](int g, int64_t off, int64_t cnt) {
            hw_zbc_request q = r;
            q.n_pairs = cnt;
            q.path_offset = base_off + (uint64_t)off;
            rc[g] = hw_zbc_moments(ctx[g], &q, &part[5 * g], &t[g]);
            if (rc[g] != HW_OK) err[g] = hw_last_error();          // thread-local: read it on the thread that failed
        });
        for (size_t g = 0; g < G; g++) if (rc[g] != HW_OK) die("hw_zbc_moments", err[g].c_str());
        for (int k = 0; k < 5; k++) { out[k] = 0.0; for (size_t g = 0; g < G; g++) out[k] += part[5 * g + k]; }
        if (ms) *ms = *std::max_element(t.begin(), t.end());
    }
    void zcb(int64_t n, uint64_t seed, double sigma, std::vector<double>* P_sum, float* ms)
    {
        if (comm) { P_sum->assign(p.n_mat, 0.0); HW_CHECK(hw_comm_zcb_curve(comm, n, seed, sigma, P_sum->data(), ms)); return; }
        const size_t G = ctx.size();
        std::vector<double> part((size_t)p.n_mat * G, 0.0);
        std::vector<float> t(G, 0.0f);
        std::vector<int> rc(G, 0);
        std::vector<std::string> err(G);
        for_each_device(n, [&](int g, int64_t off, int64_t cnt) {
            rc[g] = hw_zcb_curve(ctx[g], cnt, seed, (uint64_t)off, sigma, &part[(size_t)p.n_mat * g], &t[g]);
            if (rc[g] != HW_OK) err[g] = hw_last_error();
        });
        for (size_t g = 0; g < G; g++) if (rc[g] != HW_OK) die("hw_zcb_curve", err[g].c_str());
        P_sum->assign(p.n_mat, 0.0);
        for (int j = 0; j < p.n_mat; j++) for (size_t g = 0; g < G; g++) (*P_sum)[j] += part[(size_t)p.n_mat * g + j];
        if (ms) *ms = *std::max_element(t.begin(), t.end());
    }
    double vega(int64_t n, uint64_t seed, double S1, double S2, double K, float* ms)
    {
        if (comm) { double t = 0.0; HW_CHECK(hw_comm_vega_pathwise(comm, n, seed, S1, S2, K, 0, &t, ms)); return t; }
        const size_t G = ctx.size();
        std::vector<double> part(G, 0.0);
        std::vector<float> t(G, 0.0f);
        std::vector<int> rc(G, 0);
        std::vector<std::string> err(G);
        for_each_device(n, [&](int g, int64_t off, int64_t cnt) {
            rc[g] = hw_vega_pathwise(ctx[g], cnt, seed, (uint64_t)off, S1, S2, K, 0, &part[g], &t[g]);
            if (rc[g] != HW_OK) err[g] = hw_last_error();
        });
        double total = 0.0;
        for (size_t g = 0; g < G; g++) { if (rc[g] != HW_OK) die("hw_vega_pathwise", err[g].c_str()); total += part[g]; }
        if (ms) *ms = *std::max_element(t.begin(), t.end());
        return total;
    }
};

// control-variate post-processing (2_option:154-175) -> named fields
struct CvResult { double mean_X, mean_Y, var_Y, cov, beta, adjustment, price, var_X, rho; };
inline CvResult cv_price(const double sums[5], int64_t n_total, double P0S2)
{
    double r[9];
    HW_CHECK(hw_cv_price(sums, n_total, P0S2, r));
    return {r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8]};
}

// 20-run statistics: 2_option:305-324, 3_sens:570-589 (t critical 2.093 for df = 19)
struct RunStats { double mean, var, sd, se, margin, lo, hi, cv_percent; };
inline RunStats run_stats(const std::vector<double>& x)
{
    RunStats s{};
    const int n = (int)x.size();
    for (double v : x) s.mean += v;
    s.mean /= n;
    for (double v : x) s.var += (v - s.mean) * (v - s.mean);
    s.var /= (n - 1);
    s.sd = std::sqrt(s.var);
    s.se = s.sd / std::sqrt((double)n);
    s.margin = 2.093 * s.se;
    s.lo = s.mean - s.margin;
    s.hi = s.mean + s.margin;
    s.cv_percent = 100.0 * s.sd / s.mean;
    return s;
}

inline void load_market(const Options& o, const hw_params& p, std::vector<float>* P, std::vector<float>* f)
{
    P->assign(p.n_mat, 0.0f);
    f->assign(p.n_mat, 0.0f);
    if (o.market == "exact") {
        HW_CHECK(hw_analytic_curve(&p, P->data(), f->data()));
        printf("Market curve: exact Hull-White P(0,T), f(0,T) for theta profile '%s'\n", o.profile.c_str());
    } else {
        load_array("data/P.bin", P->data(), p.n_mat);      // P_FILE / F_FILE, common.cuh:43-44
        load_array("data/f.bin", f->data(), p.n_mat);
    }
}

}  // namespace app
