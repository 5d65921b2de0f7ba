// benchmark_reductions.hip -- the reference's optional reduction-strategy benchmark (src/benchmark_reductions.cu:
// the SAME ZBC path kernel finished by naive atomics / a shared-memory tree / warp shuffles, 2 warm-ups + 5 timed
// runs, data/benchmark_reductions.json) re-done for MI355X.  The path loop is the production one (ou_walk, Philox
// and Box-Muller from csrc/hw_kernels.hpp); only the way the five moment sums leave the threads differs:
//
//   naive          every thread: 5 atomicAdd(double) to global memory          (benchmark_reductions.cu naive)
//   lds_tree       workgroup LDS tree, thread 0: 5 atomics                     (... shared_memory)
//   wave_shuffle   __shfl_down (ds_bpermute) per wave, lane 0: 5 atomics       (... warp_optimized)
//   dpp_partials   DPP wave sums -> LDS -> one row of partials per workgroup -> fixed-order column pass
//                  (the library's default: no atomics, bit-reproducible)
//   dpp_ticket     the same finished inside the launch by the last workgroup (sharded tickets)
//
// Reads data/P.bin, data/f.bin like the reference's benchmark ("Did you run Q1 first?", common.cuh:155-161).
//   make benchmark && ./bin/benchmark [--paths-log2 20] [--seed 1234] [--theta-profile head|readme]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/hw_mc.h"
#include "../monte-carlo-simulation-of-hull-white-model-and-sensitivities-computation_amd/csrc/hw_kernels.hpp"
#include "../monte-carlo-simulation-of-hull-white-model-and-sensitivities-computation_amd/csrc/hw_model.hpp"

#define CHECK(x)                                                                                        \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) { printf("HIP Error [%s]: %s\n", #x, hipGetErrorString(e_)); exit(1); }   \
    } while (0)

enum Mode { NAIVE = 0, LDS_TREE, WAVE_SHUFFLE, DPP_PARTIALS, DPP_TICKET };

template <int MODE>
__global__ void __launch_bounds__(1024)
zbc_bench_kernel(const hwk::ZbcArgs<float, 1> a, double* __restrict__ acc5, double* __restrict__ partials,
                 unsigned* __restrict__ ticket, double* __restrict__ out)
{
    using namespace hwk;
    __shared__ double smem[1024 * 5];                 // LDS_TREE needs one slot per thread and sum
    const PhiloxKey key = philox_key(a.seed);
    PhiloxKey vkey = key;
    philox_key_to_vgprs(vkey);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = p < a.n_pairs;
    PhiloxPath pp[1] = {philox_path(key, a.path_offset + (uint64_t)(valid ? p : 0))};
    float y[1] = {0.0f}, S[1] = {0.0f};
    ou_walk<float, 1>(vkey, pp, a.n_steps, a.e, y, S);
    const ZbcScen<float>& sc = a.sc[0];
    const float h = S[0] - 0.5f * y[0];
    const float z = sc.cz * h, w = sc.cw * y[0];
    const float d1 = sc.D0 * hw_exp2(-z), d2 = sc.D0 * hw_exp2(z);
    const float P1 = sc.P0 * hw_exp2(-w), P2 = sc.P0 * hw_exp2(w);
    const float c1 = d1 * P1, c2 = d2 * P2;
    const float x1 = d1 * hw_max0(P1 - sc.K), x2 = d2 * hw_max0(P2 - sc.K);
    double v[5] = {0, 0, 0, 0, 0};
    if (valid) {
        v[0] = (double)(x1 + x2);
        v[1] = (double)(c1 + c2);
        v[2] = (double)(x1 * x1 + x2 * x2);
        v[3] = (double)(c1 * c1 + c2 * c2);
        v[4] = (double)(x1 * c1 + x2 * c2);
    }
    if (MODE == NAIVE) {
#pragma unroll
        for (int k = 0; k < 5; k++) atomicAdd(&acc5[k], v[k]);
    } else if (MODE == LDS_TREE) {
#pragma unroll
        for (int k = 0; k < 5; k++) smem[k * blockDim.x + threadIdx.x] = v[k];
        __syncthreads();
        for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
#pragma unroll
                for (int k = 0; k < 5; k++) smem[k * blockDim.x + threadIdx.x] += smem[k * blockDim.x + threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x < 5) atomicAdd(&acc5[threadIdx.x], smem[threadIdx.x * blockDim.x]);
    } else if (MODE == WAVE_SHUFFLE) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const double s = wave_sum(v[k]);          // __shfl_down tree (ds_bpermute)
            if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&acc5[k], s);
        }
    } else {
        grid_sum_finish<5>(v, smem, partials, MODE == DPP_TICKET ? ticket : nullptr, out);
    }
}

struct Result { const char* method; float time_ms, throughput; double price; };

int main(int argc, char** argv)
{
    int paths_log2 = 20;
    unsigned long long seed = 1234;
    std::string profile = "head";
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--paths-log2" && i + 1 < argc) paths_log2 = atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 0);
        else if (a == "--theta-profile" && i + 1 < argc) profile = argv[++i];
        else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    const int64_t n = (int64_t)1 << paths_log2;
    hw_params p;
    if (hw_default_params(&p, profile.c_str()) != HW_OK) { printf("Error: %s\n", hw_last_error()); return 1; }
    std::vector<float> P(p.n_mat), f(p.n_mat);
    for (int k = 0; k < 2; k++) {                                           // load_market_data, common.cuh:144-169
        const char* path = k ? "data/f.bin" : "data/P.bin";
        FILE* fh = fopen(path, "rb");
        if (!fh) { printf("Error: Cannot open %s\nDid you run Q1 first?\n", path); return 1; }
        const size_t got = fread(k ? f.data() : P.data(), sizeof(float), p.n_mat, fh);
        fclose(fh);
        if ((int)got != p.n_mat) { printf("Error: %s is too short\n", path); return 1; }
    }
    const double S1 = 5.0, S2 = 10.0, K = (double)expf(-0.1f);              // benchmark_reductions.cu:100-102
    // scenario constants exactly as the library builds them (csrc/hw_engine.hip: build_scenario, make_scen)
    const hw::StepConstants k = hw::step_constants(p, p.sigma);
    std::vector<double> drift, m, cI;
    hw::drift_tables(p, p.sigma, &drift, nullptr);
    const int n_steps = hw::steps_to(p, S1);
    hw::mean_path(p.r0, k.e, k.dt, drift, n_steps, &m, &cI);
    double A, B;
    hw::bond_AB(p, p.sigma, P, f, S1, S2, &A, &B);
    const double log2e = 1.4426950408889634074;
    hwk::ZbcArgs<float, 1> a{};
    a.n_pairs = n;
    a.seed = seed;
    a.path_offset = 0;
    a.n_steps = n_steps;
    a.iters = 1;
    a.e = (float)k.e;
    a.sc[0].cz = (float)(k.dt * k.sig_st * log2e * hwk::kBmScale);
    a.sc[0].cw = (float)(B * k.sig_st * log2e * hwk::kBmScale);
    a.sc[0].D0 = (float)std::exp(-cI[n_steps]);
    a.sc[0].P0 = (float)(A * std::exp(-B * m[n_steps]));
    a.sc[0].K = (float)K;
    a.clk = nullptr;
    a.seed_stride = 0;

    const int block = 512, grid = (int)((n + block - 1) / block);
    a.blocks_per_run = (uint32_t)grid;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    double *d_acc, *d_partials, *d_out;
    unsigned* d_ticket;
    CHECK(hipMalloc(&d_acc, 5 * sizeof(double)));
    CHECK(hipMalloc(&d_out, 5 * sizeof(double)));
    CHECK(hipMalloc(&d_partials, (size_t)grid * 5 * sizeof(double)));
    CHECK(hipMalloc(&d_ticket, 9 * hwk::kTicketStride * sizeof(unsigned)));
    CHECK(hipMemset(d_ticket, 0, 9 * hwk::kTicketStride * sizeof(unsigned)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    printf("REDUCTION METHODS BENCHMARK (%s %s, %d CUs)\n\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    printf("Test Parameters:\n  Option: ZBC(S1=%.1f, S2=%.1f, K=%.6f)\n", S1, S2, K);
    printf("  Paths: %lld (x2 antithetic = %lld effective)\n", (long long)n, (long long)(2 * n));
    printf("  Block config: %d threads/block, %d blocks\n", block, grid);
    printf("  Number of benchmark runs: 5 (average taken), after a clock spin-up\n\nRunning benchmarks...\n\n");

    auto launch = [&](int mode) {
        CHECK(hipMemsetAsync(d_acc, 0, 5 * sizeof(double), 0));
        switch (mode) {
            case NAIVE: hipLaunchKernelGGL(zbc_bench_kernel<NAIVE>, dim3(grid), dim3(block), 0, 0, a, d_acc, d_partials, d_ticket, d_out); break;
            case LDS_TREE: hipLaunchKernelGGL(zbc_bench_kernel<LDS_TREE>, dim3(grid), dim3(block), 0, 0, a, d_acc, d_partials, d_ticket, d_out); break;
            case WAVE_SHUFFLE: hipLaunchKernelGGL(zbc_bench_kernel<WAVE_SHUFFLE>, dim3(grid), dim3(block), 0, 0, a, d_acc, d_partials, d_ticket, d_out); break;
            case DPP_PARTIALS:
                hipLaunchKernelGGL(zbc_bench_kernel<DPP_PARTIALS>, dim3(grid), dim3(block), 0, 0, a, d_acc, d_partials, d_ticket, d_out);
                hipLaunchKernelGGL(hwk::finalize_columns_kernel, dim3(5), dim3(256), 0, 0, (const double*)d_partials, grid, 5, 1,
                                   (const double*)nullptr, -1.0, d_out, 5);
                break;
            default: hipLaunchKernelGGL(zbc_bench_kernel<DPP_TICKET>, dim3(grid), dim3(block), 0, 0, a, d_acc, d_partials, d_ticket, d_out); break;
        }
        CHECK(hipGetLastError());
    };
    const auto t0 = std::chrono::steady_clock::now();                         // the chip's clock ramps over ~100 ms of load
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 200.0) {
        launch(DPP_PARTIALS);
        CHECK(hipDeviceSynchronize());
    }
    const char* names[] = {"Naive (direct atomicAdd f64)", "LDS tree + block atomics", "Wave shuffle + wave atomics",
                           "DPP + partial rows (default)", "DPP + in-launch tickets"};
    std::vector<Result> results;
    for (int mode = 0; mode < 5; mode++) {
        for (int wu = 0; wu < 2; wu++) launch(mode);                          // benchmark_reductions.cu: 2 warm-ups
        CHECK(hipDeviceSynchronize());
        float total = 0.0f;
        double sums[5];
        for (int run = 0; run < 5; run++) {                                   // ... 5 timed runs
            CHECK(hipEventRecord(e0, 0));
            launch(mode);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
        }
        CHECK(hipMemcpy(sums, mode >= DPP_PARTIALS ? d_out : d_acc, sizeof sums, hipMemcpyDeviceToHost));
        double res[9];
        if (hw_cv_price(sums, 2 * n, (double)P[p.n_mat - 1], res) != HW_OK) { printf("Error: %s\n", hw_last_error()); return 1; }
        const float ms = total / 5.0f;
        results.push_back({names[mode], ms, (float)((2.0 * n / ms) / 1000.0), res[6]});
        printf("  %-30s %8.3f ms   price %.8f\n", names[mode], ms, res[6]);
    }

    printf("\n================================================================================\nBENCHMARK SUMMARY\n");
    printf("================================================================================\n\n");
    printf("%-30s | %10s | %15s\n", "Method", "Time (ms)", "Throughput (M/s)");
    printf("%-30s-+-----------+-----------------\n", "------------------------------");
    for (const Result& r : results)
        printf("%-30s | %10.3f | %15.2f  (%.2fx)\n", r.method, r.time_ms, r.throughput, results[0].time_ms / r.time_ms);
    printf("\n================================================================================\nVALIDATION\n");
    printf("================================================================================\n\nPrice consistency:\n");
    double worst = 0.0;
    for (size_t i = 1; i < results.size(); i++) {
        const double d = std::fabs(results[0].price - results[i].price);
        worst = std::max(worst, d);
        printf("  Naive vs %-28s %.2e (relative: %.6f%%)\n", results[i].method, d, 100.0 * d / results[0].price);
    }
    if (worst < 1e-9) printf("\nAll methods produce identical results (within numerical precision)\n");

    FILE* json = fopen("data/benchmark_reductions.json", "w");               // schema of benchmark_reductions.cu:172-197
    if (json) {
        fprintf(json, "{\n  \"benchmark\": \"Reduction Methods Performance\",\n  \"parameters\": {\n");
        fprintf(json, "    \"N_PATHS\": %lld,\n    \"NTPB\": %d,\n    \"NB\": %d,\n    \"S1\": %.1f,\n    \"S2\": %.1f,\n    \"K\": %.6f\n  },\n",
                (long long)n, block, grid, S1, S2, K);
        fprintf(json, "  \"results\": [\n");
        for (size_t i = 0; i < results.size(); i++)
            fprintf(json, "    {\n      \"method\": \"%s\",\n      \"time_ms\": %.3f,\n      \"throughput_Mpaths_per_sec\": %.2f,\n      \"price\": %.8f\n    }%s\n",
                    results[i].method, results[i].time_ms, results[i].throughput, results[i].price, i + 1 < results.size() ? "," : "");
        fprintf(json, "  ]\n}\n");
        fclose(json);
        printf("\nSaved data/benchmark_reductions.json\n");
    }
    return 0;
}
