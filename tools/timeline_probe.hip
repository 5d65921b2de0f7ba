// timeline_probe.hip -- when do the workgroups of one path-kernel launch start and finish?
// The production loop (ou_walk, 500 steps) with an s_memrealtime stamp (100 MHz) at the entry and exit of every
// workgroup.  Prints how long the dispatcher takes to fill the chip, when the first and the last workgroup of each
// round retire, and how much of the launch is drain (the time after the last workgroup STARTED during which slots
// run empty) -- the anatomy of the fixed ~17 us per launch that tools/size_scan.py measures.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I <package>/csrc tools/timeline_probe.hip -o tools/bin/timeline_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "hw_kernels.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// The PRODUCTION kernels are launched (hwk::zbc_superposed_kernel<float,1,ILP> or hwk::zcb_curve_blocked_kernel<P>) with
// their optional `stamps` argument set; the engine itself always passes nullptr there.
//   timeline_probe [log2 pairs = 20] [threads per workgroup = 512] [kernel = zbc | zbc1 | zbc2 | zcb1 | zcb2 | zcb4] [n_mat = 101]
// zbc: pairs per thread as the engine chooses them (two from two chip-fills of pairs up); zbc1 / zbc2 force one / two.
int main(int argc, char** argv)
{
    using namespace hwk;
    const int log2n = argc > 1 ? atoi(argv[1]) : 20, block = argc > 2 ? atoi(argv[2]) : 512;
    const std::string kern = argc > 3 ? argv[3] : "zbc";
    const int n_mat = argc > 4 ? atoi(argv[4]) : 101;
    const long long n_pairs = 1ll << log2n;
    const bool is_zbc = kern.rfind("zbc", 0) == 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ilp = kern == "zbc2" ? 2 : kern == "zbc1" ? 1 : (n_pairs >= 2ll * 2048 * prop.multiProcessorCount ? 2 : 1);   // effective_ilp
    const int P = kern == "zcb2" ? 2 : kern == "zcb4" ? 4 : is_zbc ? ilp : 1;
    const int grid = (int)(n_pairs / ((long long)block * P));
    const int n_steps = is_zbc ? 500 : 1000;
    unsigned long long* d_st;
    double *d_partials, *d_out;
    CHECK(hipMalloc(&d_st, (size_t)grid * 16));
    CHECK(hipMalloc(&d_partials, (size_t)grid * (is_zbc ? 5 : n_mat) * sizeof(double)));
    CHECK(hipMalloc(&d_out, 1024 * sizeof(double)));
    const float e = 0.99004983f;
    ZbcArgs<float, 1> za{};
    za.n_pairs = n_pairs; za.seed = 1; za.n_steps = n_steps; za.iters = 1; za.e = e;
    za.sc[0].cz = 1.7e-4f; za.sc[0].cw = 1.67e-2f; za.sc[0].D0 = 0.93f; za.sc[0].P0 = 0.9f; za.sc[0].K = 0.9048374f;
    za.blocks_per_run = (uint32_t)grid;
    ZcbBlockedArgs ca{};
    ca.n_pairs = n_pairs; ca.seed = 1; ca.n_steps = n_steps; ca.n_mat = n_mat; ca.stride = n_steps / (n_mat - 1); ca.iters = 1;
    ca.e = e; ca.e2 = e * e; ca.c1 = 1.7e-2f; ca.c2 = 1.7e-2f;
    const size_t lds = (size_t)(block / 64) * maturity_groups(n_mat) * kWave * sizeof(double);
    auto launch = [&](unsigned long long seed, unsigned long long* stamps) {
        if (is_zbc) {
            za.seed = seed; za.stamps = stamps;
            if (ilp == 2) hipLaunchKernelGGL((zbc_superposed_kernel<float, 1, 2>), dim3(grid), dim3(block), 0, 0, za, d_partials, (unsigned*)nullptr, d_out);
            else hipLaunchKernelGGL((zbc_superposed_kernel<float, 1, 1>), dim3(grid), dim3(block), 0, 0, za, d_partials, (unsigned*)nullptr, d_out);
        } else {
            ca.seed = seed; ca.stamps = stamps;
            if (P == 4) hipLaunchKernelGGL(zcb_curve_blocked_kernel<4>, dim3(grid), dim3(block), lds, 0, ca, d_partials);
            else if (P == 2) hipLaunchKernelGGL(zcb_curve_blocked_kernel<2>, dim3(grid), dim3(block), lds, 0, ca, d_partials);
            else hipLaunchKernelGGL(zcb_curve_blocked_kernel<1>, dim3(grid), dim3(block), lds, 0, ca, d_partials);
        }
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < (is_zbc ? 600 : 300); i++) launch(1ull, nullptr);   // clocks up
    CHECK(hipDeviceSynchronize());
    float ms_plain = 1e9f, ms = 0.0f;
    for (int i = 0; i < 5; i++) {                                                  // the same launch without stamps
        CHECK(hipEventRecord(e0));
        launch(2ull, nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms_plain = std::min(ms_plain, t);
    }
    CHECK(hipEventRecord(e0));
    launch(2ull, d_st);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    std::vector<unsigned long long> st(2 * (size_t)grid);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long base = ~0ull;
    for (int b = 0; b < grid; b++) base = std::min(base, st[2 * b]);
    std::vector<double> start(grid), end(grid), life(grid);
    for (int b = 0; b < grid; b++) { start[b] = (st[2 * b] - base) * 0.01; end[b] = (st[2 * b + 1] - base) * 0.01; life[b] = end[b] - start[b]; }
    std::vector<double> s2 = start, e2 = end, l2 = life;
    std::sort(s2.begin(), s2.end()); std::sort(e2.begin(), e2.end()); std::sort(l2.begin(), l2.end());
    auto pct = [&](const std::vector<double>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
    printf("%s (%d pair(s) per thread): 2^%d pairs, %d-thread workgroups (%d), %d steps, %d maturities; hipEvent time of the launch %.1f us (%.1f without stamps); "
           "in-kernel span first start -> last end %.1f us (stamps: us after the first workgroup started)\n",
           kern.c_str(), P, log2n, block, grid, n_steps, is_zbc ? 0 : n_mat - 1, ms * 1e3, ms_plain * 1e3, e2.back());
    printf("workgroup START : min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", s2[0], pct(s2, .1), pct(s2, .5), pct(s2, .9), s2.back());
    printf("workgroup END   : min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", e2[0], pct(e2, .1), pct(e2, .5), pct(e2, .9), e2.back());
    printf("workgroup LIFE  : min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", l2[0], pct(l2, .1), pct(l2, .5), pct(l2, .9), l2.back());
    int first_round = 0;
    for (int b = 0; b < grid; b++) first_round += start[b] < e2[0];
    printf("workgroups started before the first one finished (first round): %d; it took %.1f us to start them\n", first_round, s2[std::max(0, first_round - 1)]);
    printf("drain: last workgroup started at %.1f us, launch ended at %.1f us -> %.1f us with no refill; "
           "workgroups alive in the last 5 / 10 / 20 us: %d / %d / %d of %d slots\n", s2.back(), e2.back(), e2.back() - s2.back(),
           (int)(e2.end() - std::lower_bound(e2.begin(), e2.end(), e2.back() - 5.0)), (int)(e2.end() - std::lower_bound(e2.begin(), e2.end(), e2.back() - 10.0)),
           (int)(e2.end() - std::lower_bound(e2.begin(), e2.end(), e2.back() - 20.0)), first_round);
    printf("workgroups alive at 5 %% steps of the span:");
    for (int k = 0; k <= 20; k++) {
        const double t = e2.back() * k / 20.0;
        int alive = 0;
        for (int b = 0; b < grid; b++) alive += start[b] <= t && end[b] > t;
        printf(" %d", alive);
    }
    printf("\n");
    return 0;
}
