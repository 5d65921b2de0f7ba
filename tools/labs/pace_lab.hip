// pace_lab.hip -- can the waves of a SIMD be made to progress at the same rate?  The issue arbiter serves old waves first
// (tools/mix_bound.hip: the oldest wave of a SIMD makes 11 000 trips through the production loop while the youngest makes
// 16), which is what turns the end of a launch into a thin, slow tail.  Here the PRODUCTION loop (REAL1 of
// tools/mix_bound_body.inc, verbatim) runs with a pacing epilogue: every trip the wave compares its own s_memtime with the
// schedule `trips x pace` and sleeps (s_sleep: issues nothing, the other waves take the slots) when it is ahead by more than
// a slack.  pace = cycles per block per SIMD x waves per SIMD x (1 + margin).  Same chip-wide deadline as mix_bound.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pace_lab.hip -o tools/bin/pace_lab ; tools/bin/pace_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../mix_bound_body.inc"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define LAB_PROLOGUE \
    "s_mov_b32 s38, 0\n\ts_mov_b32 s89, 0\n\t" \
    "s_memrealtime s[80:81]\n\ts_memtime s[76:77]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "1:\n\t"
// PACED = 1: s_memtime issued at the top of the trip, consumed at its end
#define LAB_TOP_PACED "s_memtime s[86:87]\n\t"
#define LAB_PACE \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_sub_u32 s88, s86, s76\n\t"            /* elapsed since this wave started (low 32 bits) */ \
    "s_add_u32 s89, s89, %[pace]\n\t"        /* schedule: trips x pace */ \
    "s_sub_u32 s88, s89, s88\n\t"            /* ahead = schedule - elapsed */ \
    "s_cmp_gt_i32 s88, %[slack]\n\t" \
    "s_cbranch_scc0 2f\n\t" \
    "s_sleep 8\n\t" \
    "2:\n\t"
#define LAB_EPILOGUE \
    "s_add_u32 s38, s38, 1\n\ts_and_b32 s39, s38, 15\n\ts_cmp_lg_u32 s39, 0\n\ts_cbranch_scc1 1b\n\t" \
    "s_memrealtime s[82:83]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "s_sub_u32 s84, s82, %[dlo]\n\ts_subb_u32 s85, s83, %[dhi]\n\t" \
    "s_cmp_lt_i32 s85, 0\n\ts_cbranch_scc1 1b\n\t" \
    "s_setprio 0\n\t" \
    "s_memtime s[78:79]\n\ts_memrealtime s[82:83]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "s_sub_u32 %[clo], s78, s76\n\ts_mov_b32 %[chi], s38\n\t" \
    "s_mov_b32 %[rlo], s80\n\ts_mov_b32 %[rhi], s81\n\ts_mov_b32 %[elo], s82\n\ts_mov_b32 %[ehi], s83\n\t"
#define LAB_CLOBBERS "s38", "s39", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89"

#define LAB_KERNEL(NAME, TOP, PACE)                                                                                     \
    __global__ void __launch_bounds__(512) NAME(unsigned long long* out, unsigned dlo, unsigned dhi, int pace, int slack) \
    {                                                                                                                   \
        unsigned clo, chi, rlo, rhi, elo, ehi;                                                                          \
        asm volatile(LAB_PROLOGUE TOP MIX_BODY_REAL1 PACE LAB_EPILOGUE                                                  \
                     : [clo] "=s"(clo), [chi] "=s"(chi), [rlo] "=s"(rlo), [rhi] "=s"(rhi), [elo] "=s"(elo), [ehi] "=s"(ehi) \
                     : [dlo] "s"(dlo), [dhi] "s"(dhi), [pace] "s"(pace), [slack] "s"(slack)                            \
                     : LAB_CLOBBERS, MIX_CLOBBERS_REAL1);                                                               \
        if ((threadIdx.x & 63) == 0) {                                                                                  \
            const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);                               \
            out[3 * w] = ((unsigned long long)chi << 32) | clo;                                                         \
            out[3 * w + 1] = ((unsigned long long)rhi << 32) | rlo;                                                     \
            out[3 * w + 2] = ((unsigned long long)ehi << 32) | elo;                                                     \
        }                                                                                                               \
    }
// the same, checked every 4th trip only: the s_memtime is issued in the trip before the check
#define LAB_TOP_PACED4 \
    "s_and_b32 s39, s38, 3\n\ts_cmp_lg_u32 s39, 2\n\ts_cbranch_scc1 3f\n\ts_memtime s[86:87]\n\t3:\n\t"
#define LAB_PACE4 \
    "s_add_u32 s89, s89, %[pace]\n\t" \
    "s_and_b32 s39, s38, 3\n\ts_cmp_lg_u32 s39, 3\n\ts_cbranch_scc1 2f\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_sub_u32 s88, s86, s76\n\t" \
    "s_sub_u32 s88, s89, s88\n\t" \
    "s_cmp_gt_i32 s88, %[slack]\n\t" \
    "s_cbranch_scc0 2f\n\t" \
    "s_sleep 8\n\t" \
    "2:\n\t"
// only the clock read and the wait for it, no comparison: what does reading the clock cost?
#define LAB_PACE_READONLY "s_waitcnt lgkmcnt(0)\n\t"
// ROTATING PRIORITIES: no wave sleeps; instead every wave runs the loop on priority levels 2/3 during the phases of its
// OWN progress in which ((trips >> shift) ^ group) is odd, and on the production levels 0/1 otherwise.  group = one bit per
// wave (mode picks which).  A wave that is ahead reaches its low phase sooner and yields to the others.
#define LAB_KERNEL_ROT(NAME)                                                                                            \
    __global__ void __launch_bounds__(512) NAME(unsigned long long* out, unsigned dlo, unsigned dhi, int shift, int mode) \
    {                                                                                                                   \
        const int wave = threadIdx.x >> 6;                                                                              \
        int group = mode == 0 ? (wave >> 2) : mode == 1 ? (int)blockIdx.x : mode == 2 ? ((wave >> 2) ^ (int)blockIdx.x) \
                                                                         : (int)(blockIdx.x / (gridDim.x / 2));         \
        group = __builtin_amdgcn_readfirstlane(group & 1);                                                              \
        unsigned clo, chi, rlo, rhi, elo, ehi;                                                                          \
        asm volatile(LAB_PROLOGUE                                                                                       \
                     "s_lshr_b32 s39, s38, %[shift]\n\ts_xor_b32 s39, s39, %[group]\n\ts_and_b32 s39, s39, 1\n\t"        \
                     "s_cmp_eq_u32 s39, 0\n\ts_cbranch_scc1 5f\n\t"                                                     \
                     MIX_BODY_REAL1_HI "s_branch 6f\n\t5:\n\t" MIX_BODY_REAL1 "6:\n\t" LAB_EPILOGUE                    \
                     : [clo] "=s"(clo), [chi] "=s"(chi), [rlo] "=s"(rlo), [rhi] "=s"(rhi), [elo] "=s"(elo), [ehi] "=s"(ehi) \
                     : [dlo] "s"(dlo), [dhi] "s"(dhi), [shift] "s"(shift), [group] "s"(group)                          \
                     : LAB_CLOBBERS, MIX_CLOBBERS_REAL1);                                                               \
        if ((threadIdx.x & 63) == 0) {                                                                                  \
            const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);                               \
            out[3 * w] = ((unsigned long long)chi << 32) | clo;                                                         \
            out[3 * w + 1] = ((unsigned long long)rhi << 32) | rlo;                                                     \
            out[3 * w + 2] = ((unsigned long long)ehi << 32) | elo;                                                     \
        }                                                                                                               \
    }
LAB_KERNEL_ROT(lab_rotate)
LAB_KERNEL(lab_plain, "", "")
LAB_KERNEL(lab_paced, LAB_TOP_PACED, LAB_PACE)
LAB_KERNEL(lab_paced4, LAB_TOP_PACED4, LAB_PACE4)
LAB_KERNEL(lab_readonly, LAB_TOP_PACED, LAB_PACE_READONLY)
__global__ void now_kernel(unsigned long long* out) { out[0] = __builtin_amdgcn_s_memrealtime(); }

int main(int argc, char** argv)
{
    const double bound = argc > 1 ? atof(argv[1]) : 193.0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, budget = 300000;
    unsigned long long* d_out;
    CHECK(hipMalloc(&d_out, (size_t)cus * 32 * 3 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h((size_t)cus * 32 * 3);
    auto deadline = [&]() { unsigned long long now = 0; hipLaunchKernelGGL(now_kernel, dim3(1), dim3(1), 0, 0, d_out);
                            CHECK(hipMemcpy(&now, d_out, sizeof now, hipMemcpyDeviceToHost)); return now + (unsigned long long)budget + 3000ull; };
    for (int i = 0; i < 40; i++) { const unsigned long long dl = deadline(); hipLaunchKernelGGL(lab_plain, dim3(cus * 3), dim3(512), 0, 0, d_out, (unsigned)dl, (unsigned)(dl >> 32), 0, 0); }
    CHECK(hipDeviceSynchronize());
    printf("production loop verbatim, chip filled once, common deadline (3 ms); cycles per block per SIMD; trips per wave\n");
    printf("%-28s %6s %10s %9s %9s %9s %9s %9s\n", "variant", "w/SIMD", "cyc/block", "trips min", "p10", "p50", "p90", "max");
    for (int occ : {4, 6, 8})
        for (int mode = 0; mode < 8; mode++) {
            // 0 plain | 1-3 paced every trip | 4 paced, never sleeps (slack = 2^30: the instructions alone) | 5 clock read only
            // | 6-7 paced every 4th trip
            const double margin[] = {0, 0.003, 0.01, 0.003, 0.003, 0, 0.003, 0.003};
            const double slack_blocks[] = {0, 1.0, 1.0, 4.0, -1, 0, 1.0, 4.0};
            const int pace = (int)(bound * occ * (1.0 + margin[mode]) + 0.5), slack = slack_blocks[mode] < 0 ? (1 << 30) : (int)(slack_blocks[mode] * pace);
            const int grid = cus * occ / 2;
            const size_t waves = (size_t)grid * 8;
            double best = 1e30; std::vector<double> bt;
            for (int rep = 0; rep < 3; rep++) {
                const unsigned long long dl = deadline();
                auto k = mode == 0 ? lab_plain : mode == 5 ? lab_readonly : mode >= 6 ? lab_paced4 : lab_paced;
                hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d_out, (unsigned)dl, (unsigned)(dl >> 32), pace, slack);
                CHECK(hipGetLastError());
                CHECK(hipMemcpy(h.data(), d_out, waves * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double cyc = 0, real = 0, trips = 0; unsigned long long first = ~0ull, last = 0; std::vector<double> t(waves);
                for (size_t w = 0; w < waves; w++) {
                    t[w] = (double)(h[3 * w] >> 32); trips += t[w]; cyc += (double)(h[3 * w] & 0xffffffffull); real += (double)(h[3 * w + 2] - h[3 * w + 1]);
                    first = std::min(first, h[3 * w + 1]); last = std::max(last, h[3 * w + 2]);
                }
                const double mhz = cyc / real * 100.0, per = (double)cus * 4.0 * (double)(last - first) * mhz / 100.0 / trips;
                if (per < best) { best = per; std::sort(t.begin(), t.end()); bt = t; }
            }
            char name[64];
            if (mode == 0) snprintf(name, sizeof name, "plain");
            else if (mode == 4) snprintf(name, sizeof name, "paced, never sleeps");
            else if (mode == 5) snprintf(name, sizeof name, "clock read only");
            else snprintf(name, sizeof name, "paced%s %+.1f%% slack %.0f blk", mode >= 6 ? "/4" : "", 100 * margin[mode], slack_blocks[mode]);
            printf("%-28s %6d %10.2f %9.0f %9.0f %9.0f %9.0f %9.0f\n", name, occ, best, bt.front(), bt[waves / 10], bt[waves / 2], bt[waves * 9 / 10], bt.back());
        }
    printf("\nrotating priorities (levels 2/3 while ((trips >> shift) ^ group) is odd, else 0/1)\n");
    for (int occ : {4, 6, 8})
        for (int mode = 0; mode < 4; mode++)
            for (int shift : {0, 2, 4, 6}) {
                const int grid = cus * occ / 2;
                const size_t waves = (size_t)grid * 8;
                double best = 1e30; std::vector<double> bt;
                for (int rep = 0; rep < 3; rep++) {
                    const unsigned long long dl = deadline();
                    hipLaunchKernelGGL(lab_rotate, dim3(grid), dim3(512), 0, 0, d_out, (unsigned)dl, (unsigned)(dl >> 32), shift, mode);
                    CHECK(hipGetLastError());
                    CHECK(hipMemcpy(h.data(), d_out, waves * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                    double cyc = 0, real = 0, trips = 0; unsigned long long first = ~0ull, last = 0; std::vector<double> t(waves);
                    for (size_t w = 0; w < waves; w++) {
                        t[w] = (double)(h[3 * w] >> 32); trips += t[w]; cyc += (double)(h[3 * w] & 0xffffffffull); real += (double)(h[3 * w + 2] - h[3 * w + 1]);
                        first = std::min(first, h[3 * w + 1]); last = std::max(last, h[3 * w + 2]);
                    }
                    const double mhz = cyc / real * 100.0, per = (double)cus * 4.0 * (double)(last - first) * mhz / 100.0 / trips;
                    if (per < best) { best = per; std::sort(t.begin(), t.end()); bt = t; }
                }
                const char* gname[] = {"wave>>2", "block", "wave>>2 ^ block", "block half"};
                char name[64];
                snprintf(name, sizeof name, "rotate %s, %d trips", gname[mode], 1 << shift);
                printf("%-28s %6d %10.2f %9.0f %9.0f %9.0f %9.0f %9.0f\n", name, occ, best, bt.front(), bt[waves / 10], bt[waves / 2], bt[waves * 9 / 10], bt.back());
            }
    return 0;
}
