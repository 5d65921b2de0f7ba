// mix_bound.hip -- the issue bound of the production loops' instruction mixes (float ZBC/Vega loop, double ZBC loop, curve loop),
// measured on the GPU they run on.
//
// tools/gen_mix_bound.py turns the production loop (gfx950 listing of csrc/hw_engine.hip) into dependency-free twins
// (tools/mix_bound_body.inc): same opcodes, same order, same operand classes, same s_setprio choreography, but no
// instruction waits for another's result.  This harness runs every variant with the chip filled exactly once at 4, 6
// and 8 waves per SIMD (512-thread workgroups = 2 waves per SIMD each; 2, 3, 4 workgroups per CU), every wave looping
// until ONE chip-wide deadline (s_memrealtime), and reports SHADER-CLOCK cycles per Philox block per SIMD:
//     cycles per block = SIMDs x span / (total trips of all waves x blocks per trip),  span = first wave's start to last wave's end
// (s_memtime ticks at the shader clock -- the clock itself is reported from s_memrealtime's constant 100 MHz -- so the
// figure does not depend on what clock the box holds).  The BOUND is the smallest figure any dependency-free variant
// reaches at any occupancy: the production loop has the same instructions plus their dependencies, so it cannot issue
// faster.  REAL1 / REAL2 are the production loops verbatim: the calibration that this harness reproduces their rate.
//   python tools/gen_mix_bound.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mix_bound.hip -o tools/bin/mix_bound
//   tools/bin/mix_bound [window in 10 ns ticks] [--json]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#ifndef MIX_INC
#define MIX_INC "mix_bound_body.inc"        // -DMIX_INC='"mix_search_body.inc"': the random-ordering search (gen_mix_bound.py --search N)
#endif
#include MIX_INC

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// inputs v0-v7 from one per-lane float; trip counter s38; clocks in s[76:87] (<= 96 SGPRs in all: 8 waves per SIMD).
// FIXED TIME with ONE deadline for the whole chip, not fixed work: every wave loops until s_memrealtime (the chip-wide
// 100 MHz counter) passes the deadline the host computed from a reading of the same counter, checked every 16 trips, and
// reports how many trips it made.  A fixed trip count does not measure the SIMD's throughput: the issue arbiter serves
// old waves first and a dependency-free stream always has an instruction ready, so the oldest waves of a SIMD take it
// over -- the youngest make 16 trips while the oldest make 20 000 -- and a fixed job leaves the SIMD under-occupied while
// the starved waves finish (a per-wave time window has the same flaw: a starved wave STARTS its window late).  With a
// common deadline the SIMD is saturated from the first wave's start to the deadline whoever gets the slots.
#define MIX_PROLOGUE \
    "v_mov_b32 v0, %[in]\n\tv_add_f32 v1, 0.25, v0\n\tv_add_f32 v2, 0.5, v0\n\tv_add_f32 v3, 1.0, v0\n\t" \
    "v_add_f32 v4, 2.0, v0\n\tv_mul_f32 v5, 0.5, v0\n\tv_add_f32 v6, 4.0, v0\n\tv_mul_f32 v7, 2.0, v0\n\t" \
    "s_mov_b32 s38, 0\n\t" \
    "s_memrealtime s[80:81]\n\ts_memtime s[76:77]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "1:\n\t"
// MASK: the deadline is looked at every (MASK + 1) trips -- 15 for the one- and two-block loops; 0 for the curve twins, whose
// trip is a whole period of sixteen maturities (40 or 320 pair-blocks)
#define MIX_EPILOGUE(MASK) \
    "s_add_u32 s38, s38, 1\n\ts_and_b32 s39, s38, " #MASK "\n\ts_cmp_lg_u32 s39, 0\n\ts_cbranch_scc1 1b\n\t" \
    "s_memrealtime s[82:83]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "s_sub_u32 s84, s82, %[dlo]\n\ts_subb_u32 s85, s83, %[dhi]\n\t" \
    "s_cmp_lt_i32 s85, 0\n\ts_cbranch_scc1 1b\n\t" \
    "s_setprio 0\n\t" \
    "s_memtime s[78:79]\n\ts_memrealtime s[82:83]\n\ts_waitcnt lgkmcnt(0)\n\t" \
    "s_sub_u32 %[clo], s78, s76\n\ts_subb_u32 s85, s79, s77\n\ts_mov_b32 %[chi], s38\n\t" \
    "s_mov_b32 %[rlo], s80\n\ts_mov_b32 %[rhi], s81\n\ts_mov_b32 %[elo], s82\n\ts_mov_b32 %[ehi], s83\n\t"
#define MIX_HARNESS_CLOBBERS "s38", "s39", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85"

// out[3 w] = (trips << 32) | elapsed shader cycles, out[3 w + 1], out[3 w + 2] = s_memrealtime (100 MHz, one counter for the
// whole chip) when the wave started / ended
#define MIX_KERNEL(NAME, MASK)                                                                                        \
    __global__ void __launch_bounds__(512) mix_##NAME(unsigned long long* out, unsigned dlo, unsigned dhi, float seed)                \
    {                                                                                                                 \
        const float in = seed + 0.001f * (float)(threadIdx.x & 63);                                                   \
        unsigned clo, chi, rlo, rhi, elo, ehi;                                                                        \
        asm volatile(MIX_PROLOGUE MIX_BODY_##NAME MIX_EPILOGUE(MASK)                                                  \
                     : [clo] "=s"(clo), [chi] "=s"(chi), [rlo] "=s"(rlo), [rhi] "=s"(rhi), [elo] "=s"(elo), [ehi] "=s"(ehi) \
                     : [in] "v"(in), [dlo] "s"(dlo), [dhi] "s"(dhi)                                                                  \
                     : MIX_HARNESS_CLOBBERS, MIX_CLOBBERS_##NAME);                                                    \
        if ((threadIdx.x & 63) == 0) {                                                                                \
            const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);                             \
            out[3 * w] = ((unsigned long long)chi << 32) | clo;                                                       \
            out[3 * w + 1] = ((unsigned long long)rhi << 32) | rlo;                                                   \
            out[3 * w + 2] = ((unsigned long long)ehi << 32) | elo;                                                   \
        }                                                                                                             \
    }

#define MIX_DEFINE(NAME, DF, FM, FAMILY, MASK, WSCALE) MIX_KERNEL(NAME, MASK)
MIX_VARIANTS(MIX_DEFINE)

typedef void (*Kern)(unsigned long long*, unsigned, unsigned, float);

__global__ void now_kernel(unsigned long long* out) { out[0] = __builtin_amdgcn_s_memrealtime(); }
struct Variant { const char* name; Kern k; int blocks; bool dependency_free; bool full_mix; const char* family; int window_scale; };
struct Family { std::string name, bound_at; double bound = 1e30, real = 1e30; };

int main(int argc, char** argv)
{
    int budget = 300000;                                   // window in s_memrealtime ticks of 10 ns (3 ms)
    bool json = false;
    int only_occ = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--json")) json = true;
        else if (!strcmp(argv[i], "--occ") && i + 1 < argc) only_occ = atoi(argv[++i]);
        else budget = atoi(argv[i]);
    }
    if (budget < 20000) budget = 20000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
#define MIX_ROW(NAME, DF, FM, FAMILY, MASK, WSCALE) {#NAME, mix_##NAME, MIX_BLOCKS_##NAME, DF != 0, FM != 0, FAMILY, WSCALE},
    const Variant vars[] = { MIX_VARIANTS(MIX_ROW) };
    std::vector<int> occs = {4, 6, 8};                     // waves per SIMD
    if (only_occ) occs.assign(1, only_occ);
    const size_t max_waves = (size_t)cus * 4 * 8;
    unsigned long long* d_out;
    CHECK(hipMalloc(&d_out, max_waves * 3 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(max_waves * 3);
    if (!json) {
        printf("device %s CUs=%d  window=%d x 10 ns  census per block: %d single-rate / %d half-rate / %d quarter-rate VALU instructions\n",
               prop.gcnArchName, cus, budget, MIX_CENSUS_PROD1_FULL, MIX_CENSUS_PROD1_HALF, MIX_CENSUS_PROD1_QUARTER);
        printf("shader-clock cycles per Philox block per SIMD = SIMDs x span / total blocks; trips per wave min .. max; clock from s_memrealtime; span/win = (last end - first start) over the window asked for; start us = latest wave start after the first\n");
        printf("%-14s %6s %10s %12s %12s %9s %8s %9s\n", "variant", "w/SIMD", "cyc/block", "trips min", "trips max", "MHz", "span/win", "start us");
    }
    double bound = 1e30, real1 = 1e30, real2 = 1e30, excl = 1e30, simple = 1e30, mhz_sum = 0.0;
    int mhz_n = 0;
    std::string bound_at, rows;
    // warm the clocks: ~0.15 s of the production mix
    auto deadline = [&](int scale = 1) -> unsigned long long {   // the chip's own 100 MHz counter now + the window (+ 30 us for the launch)
        unsigned long long now = 0;
        hipLaunchKernelGGL(now_kernel, dim3(1), dim3(1), 0, 0, d_out);
        CHECK(hipMemcpy(&now, d_out, sizeof now, hipMemcpyDeviceToHost));
        return now + (unsigned long long)budget * (unsigned long long)scale + 3000ull;
    };
    std::vector<Family> fams;
    auto fam_of = [&](const char* nm) -> Family& {
        for (Family& f : fams) if (f.name == nm) return f;
        fams.emplace_back();
        fams.back().name = nm;
        return fams.back();
    };
    for (int i = 0; i < 60; i++) {
        const unsigned long long dl = deadline();
        hipLaunchKernelGGL(vars[0].k, dim3(cus * 3), dim3(512), 0, 0, d_out, (unsigned)dl, (unsigned)(dl >> 32), 0.3f);
    }
    CHECK(hipDeviceSynchronize());
    for (const Variant& v : vars)
        for (int occ : occs) {
            const int wg_per_cu = occ / 2;
            const int grid = cus * wg_per_cu;
            int resident = 0;
            CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, v.k, 512, 0));
            if (resident < wg_per_cu) {                      // (REAL2 holds 66 VGPRs: 7 waves per SIMD at most, as in production)
                if (!json) printf("%-14s %6d   -- only %d workgroups of 512 fit a CU\n", v.name, occ, resident);
                continue;
            }
            const size_t waves = (size_t)grid * 8;
            double best = 1e30, best_tmin = 0, best_tmax = 0, best_mhz = 0, best_ratio = 0, best_stagger = 0;
            for (int rep = 0; rep < 5; rep++) {
                const unsigned long long dl = deadline(v.window_scale);
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(512), 0, 0, d_out, (unsigned)dl, (unsigned)(dl >> 32), 0.3f + 0.01f * rep);
                CHECK(hipGetLastError());
                CHECK(hipMemcpy(h.data(), d_out, waves * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double cyc = 0, real = 0, trips = 0, tmin = 1e30, tmax = 0;
                unsigned long long first = ~0ull, last = 0, latest_start = 0;
                for (size_t w = 0; w < waves; w++) {
                    const double t = (double)(h[3 * w] >> 32), c = (double)(h[3 * w] & 0xffffffffull);
                    cyc += c; real += (double)(h[3 * w + 2] - h[3 * w + 1]); trips += t;
                    tmin = std::min(tmin, t); tmax = std::max(tmax, t);
                    first = std::min(first, h[3 * w + 1]); last = std::max(last, h[3 * w + 2]); latest_start = std::max(latest_start, h[3 * w + 1]);
                }
                const double mhz = cyc / real * 100.0;
                // the chip was busy with this kernel from the first wave's start to the last wave's end: that SPAN (100 MHz
                // ticks -> shader cycles at the clock the waves saw) is what the blocks are divided by; if the waves really
                // ran side by side for the whole window it equals the window (span_over_window ~ 1)
                const double span = (double)(last - first) * mhz / 100.0;
                const double per_block = (double)cus * 4.0 * span / (trips * v.blocks);
                if (per_block < best) {
                    best = per_block; best_tmin = tmin; best_tmax = tmax; best_mhz = mhz;
                    best_ratio = (double)(last - first) / ((double)budget * v.window_scale); best_stagger = (double)(latest_start - first) / 100.0;
                }
            }
            mhz_sum += best_mhz; mhz_n++;
            if (!json) printf("%-14s %6d %10.2f %12.0f %12.0f %9.0f %8.3f %9.1f\n", v.name, occ, best, best_tmin, best_tmax, best_mhz, best_ratio, best_stagger);
            char row[384];
            snprintf(row, sizeof row, "%s{\"variant\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_block\": %.3f, \"trips_min\": %.0f, \"trips_max\": %.0f, \"shader_mhz\": %.1f, \"span_over_window\": %.4f, \"last_start_us\": %.1f}",
                     rows.empty() ? "" : ", ", v.name, occ, best, best_tmin, best_tmax, best_mhz, best_ratio, best_stagger);
            rows += row;
            const bool zbc32 = !strcmp(v.family, "zbc_f32");
            if (zbc32 && v.dependency_free && v.full_mix && best_ratio < 1.05 /* span ~ window: the waves really ran side by side */ && best < bound) { bound = best; bound_at = std::string(v.name) + " @ " + std::to_string(occ) + " waves/SIMD"; }
            if (v.full_mix) {
                Family& f = fam_of(v.family);
                if (v.dependency_free && best_ratio < 1.05 && best < f.bound) { f.bound = best; f.bound_at = std::string(v.name) + " @ " + std::to_string(occ) + " waves/SIMD"; }
                if (!v.dependency_free) f.real = std::min(f.real, best);
            }
            if (!strcmp(v.name, "REAL1")) real1 = std::min(real1, best);
            if (!strcmp(v.name, "REAL2")) real2 = std::min(real2, best);
            if (!strcmp(v.name, "EXCL")) excl = std::min(excl, best);
            if (!strcmp(v.name, "SIMPLE")) simple = std::min(simple, best);
        }
    const int f = MIX_CENSUS_PROD1_FULL, hf = MIX_CENSUS_PROD1_HALF, q = MIX_CENSUS_PROD1_QUARTER;
    if (json) {
        printf("{\"tool\": \"tools/mix_bound.hip\", \"device\": \"%s\", \"compute_units\": %d, \"window_ticks_10ns\": %d, "
               "\"census_per_block\": {\"full\": %d, \"half\": %d, \"quarter\": %d}, "
               "\"bound_cycles_per_block\": %.3f, \"bound_at\": \"%s\", "
               "\"exclusive_classes_nominal_floor\": %d, \"exclusive_classes_measured\": %.3f, \"single_rate_only_measured\": %.3f, "
               "\"nominal_serial_model\": %d, \"real_loop_cycles_per_block\": {\"one_pair_per_thread\": %.3f, \"two_pairs_per_thread\": %.3f}, "
               "\"shader_mhz_mean\": %.1f, \"loops\": {",
               prop.gcnArchName, cus, budget, f, hf, q, bound, bound_at.c_str(), 4 * hf + 8 * q, excl, simple, 2 * f + 4 * hf + 8 * q,
               real1, real2, mhz_sum / mhz_n);
        // every production loop with a twin: its bound (fastest dependency-free full mix of the family) and, where the loop can run
        // verbatim in this harness (straight-line loops), its own rate here
        for (size_t i = 0; i < fams.size(); i++) {
            printf("%s\"%s\": {\"bound_cycles_per_block\": %.3f, \"bound_at\": \"%s\", \"real_loop_cycles_per_block\": ", i ? ", " : "",
                   fams[i].name.c_str(), fams[i].bound, fams[i].bound_at.c_str());
            if (fams[i].real < 1e29) printf("%.3f}", fams[i].real); else printf("null}");
        }
        printf("}, \"rows\": [%s]}\n", rows.c_str());
    } else {
        printf("\nBOUND (fastest dependency-free full mix): %.2f cycles per block  [%s]\n", bound, bound_at.c_str());
        printf("exclusive classes alone (16 mad + 4 cvt + 8 transcendental): nominal 4/4/8 floor %d, measured %.2f\n", 4 * hf + 8 * q, excl);
        printf("single-rate instructions alone (%d): nominal %d, measured %.2f\n", f, 2 * f, simple);
        printf("nominal serial model 2/4/8: %d\n", 2 * f + 4 * hf + 8 * q);
        printf("production loop verbatim in this harness: %.2f (one pair per thread), %.2f (two pairs)\n", real1, real2);
        printf("\nper production loop (cycles per pair-block per SIMD):\n");
        for (const Family& fm : fams) {
            printf("  %-14s bound %.2f  [%s]", fm.name.c_str(), fm.bound, fm.bound_at.c_str());
            if (fm.real < 1e29) printf("   the loop verbatim in this harness: %.2f", fm.real);
            printf("\n");
        }
    }
    CHECK(hipFree(d_out));
    return 0;
}
