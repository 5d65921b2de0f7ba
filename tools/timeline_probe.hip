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
#include <vector>
#include "hw_kernels.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) probe_kernel(int n_steps, float e, unsigned long long seed, unsigned long long* stamps, float* sink)
{
    using namespace hwk;
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    const PhiloxKey key = philox_key(seed);
    PhiloxKey vkey = key;
    philox_key_to_vgprs(vkey);
    PhiloxPath pp[1] = {philox_path(key, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x)};
    float y[1] = {0.0f}, S[1] = {0.0f};
    ou_walk<float, 1>(vkey, pp, n_steps, e, y, S);
    if (y[0] == 123.456f) sink[0] = S[0];
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 20, block = argc > 2 ? atoi(argv[2]) : 512, n_steps = argc > 3 ? atoi(argv[3]) : 500;
    const int grid = (int)(((long long)1 << log2n) / block);
    unsigned long long* d_st;
    float* d_sink;
    CHECK(hipMalloc(&d_st, (size_t)grid * 16));
    CHECK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 600; i++) hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(block), 0, 0, n_steps, 0.99f, 1ull, d_st, d_sink);   // clocks up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(block), 0, 0, n_steps, 0.99f, 2ull, d_st, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * (size_t)grid);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long base = ~0ull;
    for (int b = 0; b < grid; b++) base = std::min(base, st[2 * b]);
    std::vector<double> start(grid), end(grid), life(grid);
    for (int b = 0; b < grid; b++) { start[b] = (st[2 * b] - base) * 0.01; end[b] = (st[2 * b + 1] - base) * 0.01; life[b] = end[b] - start[b]; }
    std::vector<double> s2 = start, e2 = end, l2 = life;
    std::sort(s2.begin(), s2.end()); std::sort(e2.begin(), e2.end()); std::sort(l2.begin(), l2.end());
    auto pct = [&](const std::vector<double>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
    printf("2^%d pairs, %d-thread workgroups (%d), %d steps; hipEvent time of the launch %.1f us (stamps: us after the first workgroup started)\n",
           log2n, block, grid, n_steps, ms * 1e3);
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
    return 0;
}
