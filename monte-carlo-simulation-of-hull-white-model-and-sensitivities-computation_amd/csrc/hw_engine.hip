// hw_engine.hip -- implementation of the C ABI in include/hw_mc.h on top of the gfx950 kernels
// in hw_kernels.hpp.  Host work per launch: build the (sigma, drift-mode) tables and the mean
// path in double (hw_model.hpp), pass a handful of scalars as kernel arguments, launch the
// path kernel and the one-row-per-workgroup column reduction on the context's stream.
//
// There is NO CPU fallback in this library: every compute entry point launches HIP kernels
// and fails with HW_E_HIP if the runtime or the GPU is unavailable.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl.so.1 is dlopen'ed by hw_comm_create
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hw_mc.h"
#include "../../include/hw_mc_diag.h"
#include "hw_kernels.hpp"
#include "hw_model.hpp"
#include "hw_stream_guard.hpp"
#include "hw_fleet.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(HW_E_HIP, "HIP error [%s] at %s:%d: %s", #expr, __FILE__, __LINE__, \
                        hipGetErrorString(e_));                                           \
    } while (0)

constexpr double kLog2e = 1.4426950408889634074;

}  // namespace

// host-side constants of one (sigma, drift table, option) combination: see build_scenario()
struct Scenario {
    double sigma;
    int drift_mode;
    hw::StepConstants k;
    std::vector<double> drift;
    double m_n, cI_n;   // mean short rate and mean integral at S1
    double A, B;
    // cache key (the inputs build_scenario() was called with)
    double key_sigma, key_S1, key_S2;
    int key_drift_mode, key_steps;
};

struct hw_ctx {
    hw_params p{};
    int device = 0;
    hipDeviceProp_t prop{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<float> P, f;
    bool have_market = false;
    // device workspaces
    double* d_partials = nullptr;
    size_t partials_cap = 0;          // doubles
    double* d_out = nullptr;          // result staging, kOutCap doubles
    // The same in pinned, device-visible HOST memory: the synchronous entry points let the column pass store its few
    // doubles straight to the host (posted PCIe writes, visible after the stream synchronisation) instead of running a
    // device->host copy behind it (a blit kernel of its own: ~4 us + a launch gap per call, rocprofv3 of bin/q2).
    double* h_out = nullptr;          // host address
    double* h_out_dev = nullptr;      // the address the device uses for it
    // ... followed by kOutCap 8-byte completion stamps: the column pass of a synchronous call publishes the call's sequence
    // number behind each result it stores, and the host spins on those for a short while before it falls back to the
    // stream synchronisation (whose wake-up costs several microseconds): timed_sync
    unsigned long long* h_stamps = nullptr;      // host address
    unsigned long long* h_stamps_dev = nullptr;  // device address; nullptr outside a synchronous call
    unsigned long long stamp_seq = 0;
    unsigned long long* stamp_arm = nullptr;     // set by timed_sync for the duration of its enqueue
    int stamped = 0;                              // how many stamps the enqueue asked the column pass for (0: none, fall back)
    int spin_wait = 1;                            // env HWMC_SPIN_WAIT=0: always hipStreamSynchronize (A/B)
    double* d_scale = nullptr;        // per-column scale for the curve (n_mat doubles)
    float* d_drift = nullptr;         // float table for the reference stepper / sample paths
    unsigned* d_ticket = nullptr;     // grid_sum_finish ticket counter (self re-arming)
    unsigned long long* d_clk = nullptr;  // clock probes of the last profiled superposed-ZBC / blocked-curve launch (kClkProbes x 4 words)
    // launch knobs
    int block_size = 0;               // 0 = automatic (see geometry())
    int max_blocks_per_cu = 0;        // in units of block_size-thread workgroups; 0 = default
    int table_in_lds = 0;
    int ilp = 0;                      // pairs in flight per thread: 0 = automatic (effective_ilp), 1, 2
    int single_launch = 0;            // 0: path kernel + column pass; 1: in-kernel sharded-ticket finish
    int curve_generic = 0;            // env HWMC_CURVE_GENERIC=1: always run the per-step curve kernel (A/B tests)
    // the last few scenarios built (tables + mean path + A, B cost ~8 us of host time per pricing call otherwise);
    // emptied by hw_set_market
    std::vector<Scenario> scen_cache;
    int vega_key_steps = -1;          // mean of d_sigma r and of its integral at S1 (depend on the parameters only)
    double vega_Dn = 0.0, vega_cDn = 0.0;
    // optional per-launch event pairs around the path kernel (hw_profile_enable)
    std::vector<hipEvent_t> prof_ev;
    int prof_used = 0, prof_cap = 0;
    // The workspaces above are reused by every launch of the context, in stream order.  A call that arrives on a
    // DIFFERENT stream than the previous one first makes its stream wait (on the device) for everything the context
    // has enqueued so far, and the library never hands a caller's stream handle to HIP after the call that received it
    // has returned: hw_stream_guard.hpp (the state machine, checked on the CPU with a mock) + on_stream() below.
    hw::StreamGuard<hipStream_t> guard;
    hipEvent_t ev_tail = nullptr;         // "everything this context has enqueued so far" (see the guard)
    // workspaces outgrown while launches may still be reading them: freed at the next host synchronisation / destroy
    std::vector<void*> retired;
    // pinned staging ring for the small host->device uploads (drift table of the REFERENCE stepper, per-maturity
    // scales of the curve): a slot is reused only after the copy that last read it has completed (its event)
    struct Stage { void* h = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
    Stage stage[4];
    int stage_next = 0;
    // what d_drift / d_scale currently hold (skip the upload when a launch asks for the same table again)
    double drift_key_sigma = -1.0; int drift_key_mode = -1, drift_key_n = -1; bool drift_valid = false;
    double scale_key_sigma = -1.0; bool scale_valid = false;
};

namespace {

constexpr int kOutCap = 1024;

// Grow the partial-sum workspace.  Launches already enqueued may still be writing the old buffer, and hipFree
// synchronises the whole device, so the old buffer is only RETIRED here; free_retired() releases it once the host
// has synchronised with the context's work (the synchronous entry points, hw_destroy).
int ensure_partials(hw_ctx* c, size_t doubles)
{
    if (doubles <= c->partials_cap) return HW_OK;
    const size_t want = std::max(doubles, c->partials_cap + c->partials_cap / 2);
    double* fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, want * sizeof(double)));
    if (c->d_partials) c->retired.push_back(c->d_partials);
    c->d_partials = fresh;
    c->partials_cap = want;
    return HW_OK;
}

void free_retired(hw_ctx* c)          // caller guarantees that nothing enqueued before this call is still running
{
    for (void* p : c->retired) (void)hipFree(p);
    c->retired.clear();
}

// Stream-order guard of the per-context workspaces (d_partials, d_ticket, d_drift, d_scale): hw_stream_guard.hpp.
struct GuardOps {
    hw_ctx* c;
    hipError_t err = hipSuccess;
    const char* what = "";
    int record(hipStream_t s) { what = "hipEventRecord"; err = hipEventRecord(c->ev_tail, s); return err == hipSuccess ? 0 : 1; }
    int wait(hipStream_t s) { what = "hipStreamWaitEvent"; err = hipStreamWaitEvent(s, c->ev_tail, 0); return err == hipSuccess ? 0 : 1; }
    int sync(hipStream_t s) { what = "hipStreamSynchronize"; err = hipStreamSynchronize(s); return err == hipSuccess ? 0 : 1; }
};

int guard_result(int r, const GuardOps& ops)
{
    if (r == 0) return HW_OK;
    if (r == hw::kGuardInternalError) return fail(HW_E_HIP, "internal: unsealed caller stream in the context's stream guard");
    return fail(HW_E_HIP, "HIP error [%s] in the stream guard: %s", ops.what, hipGetErrorString(ops.err));
}

int use_stream(hw_ctx* c, hipStream_t st)
{
    GuardOps ops{c};
    return guard_result(c->guard.enter(ops, st), ops);
}

int seal_stream(hw_ctx* c, hipStream_t st)
{
    GuardOps ops{c};
    return guard_result(c->guard.leave(ops, st), ops);
}

// Every enqueue of a context runs inside this bracket (whatever it returns, the stream is sealed).
template <typename F>
int on_stream(hw_ctx* c, hipStream_t st, F body)
{
    int rc = use_stream(c, st);
    if (rc) return rc;
    rc = body();
    const int rs = seal_stream(c, st);
    return rc ? rc : rs;
}

// Copy `bytes` from host memory to `dst` on stream `st` through the pinned ring: the caller's buffer may be a
// stack/vector temporary (it is copied into pinned memory before this returns) and the call never blocks on the
// device unless all four slots still have copies in flight.
int stage_upload(hw_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    hw_ctx::Stage& sl = c->stage[c->stage_next];
    c->stage_next = (c->stage_next + 1) % 4;
    if (sl.pending) {
        HIP_TRY(hipEventSynchronize(sl.ev));
        sl.pending = false;
    }
    if (sl.cap < bytes) {
        if (sl.h) HIP_TRY(hipHostFree(sl.h));
        sl.h = nullptr;
        sl.cap = 0;
        HIP_TRY(hipHostMalloc(&sl.h, std::max<size_t>(bytes, 4096), hipHostMallocDefault));
        sl.cap = std::max<size_t>(bytes, 4096);
    }
    if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    memcpy(sl.h, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, sl.h, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(sl.ev, st));
    sl.pending = true;
    return HW_OK;
}

struct Geometry {
    int block, grid, iters;
};

// RAII-less helper: record the "before"/"after" event of the next profiled launch, if any.
inline void prof_mark(hw_ctx* c, hipStream_t st, int which)
{
    if (c->prof_used >= c->prof_cap) return;
    (void)hipEventRecord(c->prof_ev[2 * c->prof_used + which], st);
    if (which == 1) c->prof_used++;
}

// One slot = one thread-iteration handling `ilp` units.  By default every thread handles ONE slot
// (grid = all the work, up to 2^20 workgroups): with more workgroups than the chip holds at once the
// dispatcher hands them out as CUs free up, which balances the tail better than a residency-sized
// grid whose threads loop (measured at 2^20 / 2^23 pairs: 0.2245 vs 0.2313 ms, 1.698 vs 1.776 ms).
// `rows_per_block_bytes` bounds the partials buffer for kernels with wide rows (the curve): their
// grid is capped at 64 waves' worth of workgroups x 4 per CU (a 26 MB partials buffer at 101 maturities) and the
// threads loop over the rest.
Geometry geometry(const hw_ctx* c, int64_t n_units, int ilp, bool wide_rows = false)
{
    Geometry g{};
    // Workgroup size: 512 threads once that still gives every CU >= 4 workgroups (half as many rows for the
    // column pass: +1 % per pricing call at 2^20 pairs; 1024 is as good for the ZBC kernel but 10 % slower for
    // the Vega and multi-scenario kernels), else 256 so that small batches spread over the chip.
    g.block = c->block_size;
    if (g.block == 0) g.block = (n_units >= (int64_t)2048 * c->prop.multiProcessorCount * ilp) ? 512 : 256;
    const int64_t per_block = (int64_t)g.block * ilp;
    int64_t want = (n_units + per_block - 1) / per_block;
    if (want < 1) want = 1;
    const int waves_per_block = g.block / 64;
    int64_t cap;
    if (c->max_blocks_per_cu > 0) cap = (int64_t)c->prop.multiProcessorCount * c->max_blocks_per_cu;
    else if (wide_rows) cap = (int64_t)c->prop.multiProcessorCount * std::max(1, 256 / waves_per_block);
    else cap = (int64_t)1 << 20;
    g.grid = (int)std::min<int64_t>(want, cap);
    const int64_t per_iter = (int64_t)g.grid * per_block;
    g.iters = (int)((n_units + per_iter - 1) / per_iter);
    if (g.iters < 1) g.iters = 1;
    g.grid = (int)((want + g.iters - 1) / g.iters);      // same number of iterations, every one of them full
    return g;
}

// Pairs (paths) in flight per thread of the superposed ZBC / Vega kernels.  Two -- their Philox blocks interleaved
// (philox_block_prio_x2) -- halve the number of waves and with it the per-wave work outside the time loop (key setup,
// payoff epilogue, the five DPP reductions in double: ~700 issue cycles a wave).  Measured (profiles/r02_ilp_pipeline_probe.txt, one launch
// at a time): +1.0 % at 2^22 pairs and +1.2 % at 2^26 on the 1000-step grid (500 executed steps), nothing at 2000 executed
// steps (42.69 vs 42.63 ms at 2^26 pairs: the per-wave overhead is four times smaller there).  Where the rule starts was
// re-measured in the loop bench.py times (tools/labs/ilp_pipeline_probe.py, K = 20 calls, two queues / one queue): 2^20 pairs
// +1.9 % / +0.4 %, 2^21 +1.4 % / +1.0 %, 2^19 +1.1 % / -0.9 %, 2^18 -3 % / -9 % (half a chip-fill of waves).  So: two from
// TWO chip-fills of pairs up (2^20 on 256 CUs: one full round of 8 waves per SIMD) when the path has at most 1024 steps, in the
// float (y, W) loop; one for the double engine, the multi-scenario kernel and the per-step fallback under weak mean
// reversion.  hw_set_launch(pairs_in_flight = 1 | 2) overrides, -1 returns to automatic.
int effective_ilp(const hw_ctx* c, int64_t n_units, int n_steps, double e, int n_scen, int min_fills = 2)
{
    if (c->ilp == 1 || c->ilp == 2) return c->ilp;
    if (c->p.precision != HW_F32 || n_scen != 1 || n_steps > 1024) return 1;
    if ((1.0 - (double)(float)e) * std::sqrt((double)n_steps) < 0.05) return 1;       // the guard of ou_walk's (y, W) form
    return n_units >= (int64_t)min_fills * 2048 * c->prop.multiProcessorCount ? 2 : 1;
}

// Arm the clock probes of a profiled launch: zero the slots (stream order) and spread kClkProbes probing workgroups over the grid.
int arm_clock_probe(hw_ctx* c, hipStream_t st, int grid, unsigned long long** clk, uint32_t* shift)
{
    *clk = nullptr;
    *shift = 0;
    if (c->prof_cap <= 0) return HW_OK;
    HIP_TRY(hipMemsetAsync(c->d_clk, 0, hwk::kClkProbes * 4 * sizeof(unsigned long long), st));
    uint32_t sh = 0;
    while (((int64_t)hwk::kClkProbes << (sh + 1)) <= (int64_t)grid) sh++;
    *clk = c->d_clk;
    *shift = sh;
    return HW_OK;
}

// second launch of the default two-launch finish (no-op when the kernel finished in-launch)
int finish_columns(hw_ctx* c, hipStream_t st, int rows, int ncol, double* d_out, int n_groups = 1, bool finished_in_kernel = false)
{
    if (finished_in_kernel) return HW_OK;
    hipLaunchKernelGGL(hwk::finalize_columns_kernel, dim3(ncol * n_groups), dim3(256), 0, st, (const double*)c->d_partials, rows,
                       ncol, 1, (const double*)nullptr, -1.0, d_out, ncol, c->stamp_arm, c->stamp_seq);
    if (c->stamp_arm) c->stamped = ncol * n_groups;
    HIP_TRY(hipGetLastError());
    return HW_OK;
}

int build_scenario(hw_ctx* c, double sigma, int drift_mode, int n_steps_S1, double S1, double S2,
                   Scenario* s)
{
    if (!(sigma > 0.0)) sigma = c->p.sigma;
    if (drift_mode != HW_DRIFT_BASE && drift_mode != HW_DRIFT_SHIFTED)
        return fail(HW_E_INVALID, "drift_mode must be HW_DRIFT_BASE or HW_DRIFT_SHIFTED");
    for (const Scenario& q : c->scen_cache)
        if (q.key_sigma == sigma && q.key_drift_mode == drift_mode && q.key_steps == n_steps_S1 && q.key_S1 == S1 && q.key_S2 == S2) {
            *s = q;
            return HW_OK;
        }
    s->key_sigma = sigma; s->key_drift_mode = drift_mode; s->key_steps = n_steps_S1; s->key_S1 = S1; s->key_S2 = S2;
    s->sigma = sigma;
    s->drift_mode = drift_mode;
    s->k = hw::step_constants(c->p, sigma);
    if (drift_mode == HW_DRIFT_SHIFTED)
        hw::shifted_drift_table(c->p, sigma, c->p.sigma, &s->drift);
    else
        hw::drift_tables(c->p, c->p.sigma, &s->drift, nullptr);   // fitted at the base sigma
    std::vector<double> m, cI;
    hw::mean_path(c->p.r0, s->k.e, s->k.dt, s->drift, n_steps_S1, &m, &cI);
    s->m_n = m[n_steps_S1];
    s->cI_n = cI[n_steps_S1];
    hw::bond_AB(c->p, sigma, c->P, c->f, S1, S2, &s->A, &s->B);
    if (c->scen_cache.size() >= 6) c->scen_cache.erase(c->scen_cache.begin());
    c->scen_cache.push_back(*s);
    return HW_OK;
}

template <typename REAL>
hwk::ZbcScen<REAL> make_scen(const Scenario& s, double K)
{
    hwk::ZbcScen<REAL> z;
    // the kernels' y, S are the true ones divided by kBmScale: fold it into every factor that multiplies them
    z.cz = (REAL)(s.k.dt * s.k.sig_st * kLog2e * hwk::kBmScale);
    z.cw = (REAL)(s.B * s.k.sig_st * kLog2e * hwk::kBmScale);
    z.D0 = (REAL)std::exp(-s.cI_n);
    z.P0 = (REAL)(s.A * std::exp(-s.B * s.m_n));
    z.K = (REAL)K;
    return z;
}

int check_request(hw_ctx* c, const hw_zbc_request* r, int* n_steps_S1)
{
    if (!c || !r) return fail(HW_E_INVALID, "null context or request");
    if (!c->have_market) return fail(HW_E_NO_MARKET, "market curves not set (hw_set_market): did you run Q1 first?");
    if (r->n_pairs < 0) return fail(HW_E_INVALID, "n_pairs < 0");
    if (!(r->S1 > 0.0) || !(r->S2 > r->S1) || r->S2 > c->p.T_final + 1e-9)
        return fail(HW_E_INVALID, "need 0 < S1 < S2 <= T_final");
    const int n = hw::steps_to(c->p, r->S1);
    if (n < 1 || n > c->p.n_steps) return fail(HW_E_INVALID, "S1 maps to %d steps (grid has %d)", n, c->p.n_steps);
    *n_steps_S1 = n;
    return HW_OK;
}

template <typename REAL, int NSCEN>
int launch_zbc_superposed(hw_ctx* c, const hw_zbc_request* r, int n_steps_S1, const Scenario* sc,
                          double* d_out, hipStream_t st, int n_runs = 1, uint64_t seed_stride = 0)
{
    const int ilp = effective_ilp(c, r->n_pairs, n_steps_S1, sc[0].k.e, NSCEN);
    const Geometry g = geometry(c, r->n_pairs, ilp);
    if ((int64_t)g.grid * n_runs > ((int64_t)1 << 22)) return fail(HW_E_INVALID, "n_runs x workgroups per run exceeds 2^22");
    const int total_grid = g.grid * n_runs;
    int rc = ensure_partials(c, (size_t)total_grid * 5 * NSCEN);
    if (rc) return rc;
    hwk::ZbcArgs<REAL, NSCEN> a{};
    a.n_pairs = r->n_pairs;
    a.seed = r->seed;
    a.path_offset = r->path_offset;
    a.n_steps = n_steps_S1;
    a.iters = g.iters;
    a.e = (REAL)sc[0].k.e;
    for (int k = 0; k < NSCEN; k++) a.sc[k] = make_scen<REAL>(sc[k], r->K);
    rc = arm_clock_probe(c, st, total_grid, &a.clk, &a.clk_shift);
    if (rc) return rc;
    a.blocks_per_run = (uint32_t)g.grid;
    a.seed_stride = seed_stride;
    // (measured and rejected, profiles/r03_sync_call_probe.txt: finishing IN the kernel for the polled synchronous calls -- no
    //  column-pass launch -- 0.1973 against 0.1962 ms per call: the ticket finish costs the kernel more than the launch it saves)
    const bool in_kernel = n_runs == 1 && c->single_launch;
    unsigned* tk = in_kernel ? c->d_ticket : nullptr;
    prof_mark(c, st, 0);
    if (ilp == 2)
        hipLaunchKernelGGL((hwk::zbc_superposed_kernel<REAL, NSCEN, 2>), dim3(total_grid), dim3(g.block), 0, st, a, c->d_partials, tk, d_out);
    else
        hipLaunchKernelGGL((hwk::zbc_superposed_kernel<REAL, NSCEN, 1>), dim3(total_grid), dim3(g.block), 0, st, a, c->d_partials, tk, d_out);
    prof_mark(c, st, 1);
    HIP_TRY(hipGetLastError());
    return finish_columns(c, st, g.grid, 5 * NSCEN, d_out, n_runs, in_kernel);
}

int launch_zbc_reference(hw_ctx* c, const hw_zbc_request* r, int n_steps_S1, const Scenario& s,
                         double* d_out, hipStream_t st)
{
    const Geometry g = geometry(c, r->n_pairs, 1);
    int rc = ensure_partials(c, (size_t)g.grid * 5);
    if (rc) return rc;
    if (!(c->drift_valid && c->drift_key_sigma == s.sigma && c->drift_key_mode == s.drift_mode && c->drift_key_n == n_steps_S1)) {
        std::vector<float> tab(n_steps_S1);
        for (int i = 0; i < n_steps_S1; i++) tab[i] = (float)s.drift[i];
        c->drift_valid = false;
        rc = stage_upload(c, c->d_drift, tab.data(), tab.size() * sizeof(float), st);
        if (rc) return rc;
        c->drift_key_sigma = s.sigma; c->drift_key_mode = s.drift_mode; c->drift_key_n = n_steps_S1;
        c->drift_valid = true;
    }
    hwk::ZbcRefArgs a{};
    a.n_pairs = r->n_pairs;
    a.seed = r->seed;
    a.path_offset = r->path_offset;
    a.n_steps = n_steps_S1;
    a.iters = g.iters;
    a.r0 = (float)c->p.r0;
    a.e = (float)s.k.e;
    a.dt = (float)s.k.dt;
    a.sig_st = (float)(s.k.sig_st * hwk::kBmScale);       // normal4_prio yields G / kBmScale
    a.A = (float)s.A;
    a.B = (float)s.B;
    a.K = (float)r->K;
    const bool in_kernel = c->single_launch != 0;
    unsigned* tk = in_kernel ? c->d_ticket : nullptr;
    prof_mark(c, st, 0);
    if (c->table_in_lds)
        hipLaunchKernelGGL(hwk::zbc_reference_kernel<1>, dim3(g.grid), dim3(g.block),
                           (size_t)n_steps_S1 * sizeof(float), st, a, (const float*)c->d_drift, c->d_partials,
                           tk, d_out);
    else
        hipLaunchKernelGGL(hwk::zbc_reference_kernel<0>, dim3(g.grid), dim3(g.block), 0, st, a,
                           (const float*)c->d_drift, c->d_partials, tk, d_out);
    prof_mark(c, st, 1);
    HIP_TRY(hipGetLastError());
    return finish_columns(c, st, g.grid, 5, d_out, 1, in_kernel);
}

int zbc_enqueue_body(hw_ctx* c, const hw_zbc_request* r, int n_scen, const double* sigma,
                     const int32_t* drift_mode, double* d_out, hipStream_t st)
{
    int n_steps_S1 = 0;
    int rc = check_request(c, r, &n_steps_S1);
    if (rc) return rc;
    if (n_scen < 1 || n_scen > hwk::kMaxScen) return fail(HW_E_INVALID, "n_scen must be 1..%d", hwk::kMaxScen);
    Scenario sc[hwk::kMaxScen];
    for (int k = 0; k < n_scen; k++) {
        rc = build_scenario(c, sigma ? sigma[k] : r->sigma, drift_mode ? drift_mode[k] : r->drift_mode,
                            n_steps_S1, r->S1, r->S2, &sc[k]);
        if (rc) return rc;
    }
    if (r->stepper == HW_STEPPER_REFERENCE) {
        if (n_scen != 1) return fail(HW_E_INVALID, "the REFERENCE stepper prices one scenario per launch");
        if (c->p.precision != HW_F32) return fail(HW_E_INVALID, "the REFERENCE stepper is float32 only");
        return launch_zbc_reference(c, r, n_steps_S1, sc[0], d_out, st);
    }
    if (r->stepper != HW_STEPPER_SUPERPOSED) return fail(HW_E_INVALID, "unknown stepper %d", r->stepper);
    const bool f64 = c->p.precision == HW_F64;
    switch (n_scen) {
        case 1: return f64 ? launch_zbc_superposed<double, 1>(c, r, n_steps_S1, sc, d_out, st)
                           : launch_zbc_superposed<float, 1>(c, r, n_steps_S1, sc, d_out, st);
        case 2: return f64 ? launch_zbc_superposed<double, 2>(c, r, n_steps_S1, sc, d_out, st)
                           : launch_zbc_superposed<float, 2>(c, r, n_steps_S1, sc, d_out, st);
        default: return f64 ? launch_zbc_superposed<double, 3>(c, r, n_steps_S1, sc, d_out, st)
                            : launch_zbc_superposed<float, 3>(c, r, n_steps_S1, sc, d_out, st);
    }
}

int zbc_enqueue(hw_ctx* c, const hw_zbc_request* r, int n_scen, const double* sigma,
                const int32_t* drift_mode, double* d_out, hipStream_t st)
{
    return on_stream(c, st, [&] { return zbc_enqueue_body(c, r, n_scen, sigma, drift_mode, d_out, st); });
}

template <typename REAL>
int launch_zcb(hw_ctx* c, int64_t n_pairs, uint64_t seed, uint64_t path_offset, double sigma,
               double* d_P_sum, hipStream_t st)
{
    const hw_params& p = c->p;
    if (!(sigma > 0.0)) sigma = p.sigma;
    const int n_mat = p.n_mat, stride = p.n_steps / (n_mat - 1);
    const hw::StepConstants k = hw::step_constants(p, sigma);
    if (!(c->scale_valid && c->scale_key_sigma == sigma)) {
        std::vector<double> drift, m, cI;
        hw::drift_tables(p, sigma, &drift, nullptr);          // recompute_market_data, 3_sens:449-455
        hw::mean_path(p.r0, k.e, k.dt, drift, p.n_steps, &m, &cI);
        std::vector<double> scale(n_mat, 1.0);
        for (int j = 1; j < n_mat; j++) scale[j] = std::exp(-cI[(size_t)j * stride]);
        c->scale_valid = false;
        int rcs = stage_upload(c, c->d_scale, scale.data(), n_mat * sizeof(double), st);
        if (rcs) return rcs;
        c->scale_key_sigma = sigma;
        c->scale_valid = true;
    }

    // whole Philox blocks, maturities on half-block boundaries: the blocked kernel, P pairs per thread.
    // P = 2 from eight chip-fills of pairs (2^22 on 256 CUs), else 1: re-measured in round 4 after the curve loop lost its
    // two compiler artefacts (tools/labs/curve_p_now.py, profiles/r04_curve_p_scan.txt: 1000- and 4000-step grids, float and
    // double, 2^20 .. 2^26 pairs).  P = 2 halves the per-maturity reductions per pair and wins by 0.3-1.8 % from 2^22 pairs
    // up; below, the longer waves cost more in the drain than the reductions save.  P = 4 -- rounds 2-3's choice from 2^22
    // -- is never the fastest any more (2.4 % behind P = 2 at 2^22, 0.2-0.8 % at 2^26); the instance is kept for
    // HWMC_CURVE_P and the tests.
    // (in float the (y, W) form needs (1 - e) sqrt(n_steps) >= 0.05, see ou_walk; in double the division is harmless)
    constexpr bool kDouble = std::is_same<REAL, double>::value;
    if (p.n_steps % 4 == 0 && stride % 2 == 0 && stride >= 4 && !c->curve_generic &&
        (kDouble || (1.0 - (double)(float)k.e) * std::sqrt((double)p.n_steps) >= 0.05)) {
        const int64_t resident = (int64_t)c->prop.multiProcessorCount * 2048;
        int P = n_pairs >= 8 * resident ? 2 : 1;
        if (const char* m = getenv("HWMC_CURVE_P")) { const int v = atoi(m); if (v == 1 || v == 2 || v == 4) P = v; }   // tools/labs/curve_p_now.py
        const Geometry g = geometry(c, n_pairs, P, true);
        int rc = ensure_partials(c, (size_t)g.grid * n_mat);
        if (rc) return rc;
        const double inv = 1.0 / (1.0 - (double)(float)k.e), cz = k.dt * k.sig_st * kLog2e * hwk::kBmScale;
        hwk::ZcbBlockedArgs a{};
        a.n_pairs = n_pairs;
        a.seed = seed;
        a.path_offset = path_offset;
        a.n_steps = p.n_steps;
        a.n_mat = n_mat;
        a.stride = stride;
        a.iters = g.iters;
        a.e = (float)k.e;
        a.e2 = a.e * a.e;
        a.c1 = (float)(cz * inv);
        a.c2 = (float)(cz * ((double)a.e * inv + 0.5));
        a.e2_d = k.e * k.e;                                     // REAL = double: from the double e
        a.c1_d = cz / (1.0 - k.e);
        a.c2_d = cz * (k.e / (1.0 - k.e) + 0.5);
        rc = arm_clock_probe(c, st, g.grid, &a.clk, &a.clk_shift);
        if (rc) return rc;
        const size_t lds = (size_t)(g.block / 64) * hwk::maturity_groups(n_mat) * hwk::kWave * sizeof(double);
        if (lds > 64 * 1024)
            return fail(HW_E_INVALID, "n_mat = %d needs %zu B of LDS per workgroup at block size %d (limit 65536): use a smaller block", n_mat, lds, g.block);
        prof_mark(c, st, 0);
        if (P == 4) hipLaunchKernelGGL((hwk::zcb_curve_blocked_kernel<4, REAL>), dim3(g.grid), dim3(g.block), lds, st, a, c->d_partials);
        else if (P == 2) hipLaunchKernelGGL((hwk::zcb_curve_blocked_kernel<2, REAL>), dim3(g.grid), dim3(g.block), lds, st, a, c->d_partials);
        else hipLaunchKernelGGL((hwk::zcb_curve_blocked_kernel<1, REAL>), dim3(g.grid), dim3(g.block), lds, st, a, c->d_partials);
        prof_mark(c, st, 1);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(hwk::finalize_columns_kernel, dim3(n_mat), dim3(256), 0, st, (const double*)c->d_partials,
                           g.grid, n_mat, 1, (const double*)c->d_scale, 2.0 * (double)n_pairs, d_P_sum, n_mat, c->stamp_arm, c->stamp_seq);
        HIP_TRY(hipGetLastError());
        if (c->stamp_arm) c->stamped = n_mat;
        return HW_OK;
    }
    const Geometry g = geometry(c, n_pairs, 1, true);
    int rc = ensure_partials(c, (size_t)g.grid * n_mat);
    if (rc) return rc;
    hwk::ZcbArgs<REAL> a{};
    a.n_pairs = n_pairs;
    a.seed = seed;
    a.path_offset = path_offset;
    a.n_steps = p.n_steps;
    a.n_mat = n_mat;
    a.stride = stride;
    a.iters = g.iters;
    a.e = (REAL)k.e;
    a.cz = (REAL)(k.dt * k.sig_st * kLog2e * hwk::kBmScale);
    const size_t lds = (size_t)(g.block / 64) * hwk::maturity_groups(n_mat) * hwk::kWave * sizeof(double);
    if (lds > 64 * 1024)
        return fail(HW_E_INVALID, "n_mat = %d needs %zu B of LDS per workgroup at block size %d (limit 65536): use a smaller block", n_mat, lds, g.block);
    prof_mark(c, st, 0);
    hipLaunchKernelGGL(hwk::zcb_curve_kernel<REAL>, dim3(g.grid), dim3(g.block), lds, st, a, c->d_partials);
    prof_mark(c, st, 1);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(hwk::finalize_columns_kernel, dim3(n_mat), dim3(256), 0, st, (const double*)c->d_partials,
                       g.grid, n_mat, 1, (const double*)c->d_scale, 2.0 * (double)n_pairs, d_P_sum, n_mat, c->stamp_arm, c->stamp_seq);
    HIP_TRY(hipGetLastError());
    if (c->stamp_arm) c->stamped = n_mat;
    return HW_OK;
}

template <typename REAL>
int launch_vega(hw_ctx* c, int64_t n_paths, uint64_t seed, uint64_t path_offset, double S1, double S2,
                double K, int antithetic, int n_steps_S1, double* d_out, hipStream_t st, int n_runs = 1,
                uint64_t seed_stride = 0)
{
    const hw_params& p = c->p;
    Scenario s;
    int rc = build_scenario(c, p.sigma, HW_DRIFT_BASE, n_steps_S1, S1, S2, &s);
    if (rc) return rc;
    if (c->vega_key_steps != n_steps_S1) {
        std::vector<double> sdrift, D, cD;
        hw::drift_tables(p, p.sigma, nullptr, &sdrift);
        hw::mean_path(0.0, s.k.e, s.k.dt, sdrift, n_steps_S1, &D, &cD);   // 3_sens:41,44,58-61
        c->vega_Dn = D[n_steps_S1];
        c->vega_cDn = cD[n_steps_S1];
        c->vega_key_steps = n_steps_S1;
    }

    // (the Vega kernel alone on the chip at 2^20 paths: 0.1861 ms with two paths per thread against 0.1846 with one --
    // its launches are not pipelined by any caller, so it keeps the one-launch-at-a-time threshold of four chip-fills)
    const int ilp = effective_ilp(c, n_paths, n_steps_S1, s.k.e, 1, 4);
    const Geometry g = geometry(c, n_paths, ilp);
    if ((int64_t)g.grid * n_runs > ((int64_t)1 << 22)) return fail(HW_E_INVALID, "n_runs x workgroups per run exceeds 2^22");
    const int total_grid = g.grid * n_runs;
    rc = ensure_partials(c, (size_t)total_grid);
    if (rc) return rc;
    hwk::VegaArgs<REAL> a{};
    a.blocks_per_run = (uint32_t)g.grid;
    a.seed_stride = seed_stride;
    a.n_paths = n_paths;
    a.seed = seed;
    a.path_offset = path_offset;
    a.n_steps = n_steps_S1;
    a.iters = g.iters;
    a.antithetic = antithetic ? 1 : 0;
    a.e = (REAL)s.k.e;
    const hwk::ZbcScen<REAL> z = make_scen<REAL>(s, K);
    a.cz = z.cz; a.cw = z.cw; a.D0 = z.D0; a.P0 = z.P0; a.K = z.K;
    a.B = (REAL)s.B;
    a.gB = (REAL)(p.sigma / (2.0 * p.a) * (-std::expm1(-2.0 * p.a * S1)) * s.B);
    a.Dn = (REAL)c->vega_Dn;
    a.cD = (REAL)c->vega_cDn;
    a.s_over_sigma = (REAL)(s.k.sig_st / p.sigma * hwk::kBmScale);
    a.zs = (REAL)(s.k.dt * s.k.sig_st / p.sigma * hwk::kBmScale);
    const bool in_kernel = n_runs == 1 && c->single_launch;
    unsigned* tk = in_kernel ? c->d_ticket : nullptr;
    prof_mark(c, st, 0);
    if (ilp == 2)
        hipLaunchKernelGGL((hwk::vega_pathwise_kernel<REAL, 2>), dim3(total_grid), dim3(g.block), 0, st, a, c->d_partials, tk, d_out);
    else
        hipLaunchKernelGGL((hwk::vega_pathwise_kernel<REAL, 1>), dim3(total_grid), dim3(g.block), 0, st, a, c->d_partials, tk, d_out);
    prof_mark(c, st, 1);
    HIP_TRY(hipGetLastError());
    return finish_columns(c, st, g.grid, 1, d_out, n_runs, in_kernel);
}

// run `enqueue(d_out, stream)` on the context stream between events, copy n doubles back
template <typename F>
int timed_sync(hw_ctx* c, int n_out, double* out, float* kernel_ms, F enqueue)
{
    if (n_out > kOutCap) return fail(HW_E_INVALID, "too many outputs");
    HIP_TRY(hipSetDevice(c->device));
    // The event pair is recorded only for a caller who asked for the time: each record is a barrier packet of its own on
    // the queue, ~3 us of a synchronous 2^20-pair call (profiles/r03_sync_call_probe.txt).
    if (kernel_ms) HIP_TRY(hipEventRecord(c->ev0, c->stream));
    // Without the event pair the host need not wait for the runtime either: the column pass stamps every result it stores
    // into the pinned host buffer, and the call returns when all stamps of THIS call have arrived (spin for at most
    // ~300 us -- the order of the stream synchronisation's own active wait -- then fall back to hipStreamSynchronize).
    const bool spin = c->spin_wait && !kernel_ms;
    c->stamp_seq++;
    c->stamp_arm = spin ? c->h_stamps_dev : nullptr;
    c->stamped = 0;
    int rc = enqueue(c->h_out_dev, c->stream);
    c->stamp_arm = nullptr;
    if (rc) return rc;
    if (kernel_ms) HIP_TRY(hipEventRecord(c->ev1, c->stream));
    bool arrived = false;
    if (spin && c->stamped == n_out) {
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(300);
        const volatile unsigned long long* st = c->h_stamps;
        for (unsigned spins = 0;; spins++) {
            int k = 0;
            while (k < n_out && __atomic_load_n(&st[k], __ATOMIC_ACQUIRE) == c->stamp_seq) k++;
            if (k == n_out) { arrived = true; break; }
            __builtin_ia32_pause();
            if ((spins & 63) == 63 && std::chrono::steady_clock::now() > give_up) break;
        }
    }
    if (!arrived) HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_out, n_out * sizeof(double));
    // (retired workspaces: hipFree synchronises the device itself; every launch that used them precedes the column pass
    //  whose results have arrived)
    free_retired(c);
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, c->ev0, c->ev1));
    return HW_OK;
}

}  // namespace

// ==========================================================================================
extern "C" {

int hw_abi_version(void) { return HW_ABI_VERSION; }

const char* hw_last_error(void) { return g_err.c_str(); }

int hw_default_params(hw_params* p, const char* profile)
{
    if (!p) return fail(HW_E_INVALID, "null params");
    p->a = 1.0; p->sigma = 0.1; p->r0 = 0.012; p->T_final = 10.0;       // common.cuh:22,37-39
    p->n_steps = 1000; p->n_mat = 101;                                  // common.cuh:17,21
    p->th_break = 5.0; p->th_a1 = 0.012; p->th_b1 = 0.0014; p->th_b2 = 0.001;
    p->precision = HW_F32;
    p->table_mode = HW_TABLES_F64;
    if (!profile || !strcmp(profile, "head")) p->th_a2 = 0.019;         // common.cuh:76,229
    else if (!strcmp(profile, "readme")) p->th_a2 = 0.014;              // 3_sens:387; README numbers
    else return fail(HW_E_INVALID, "unknown theta profile '%s' (head|readme)", profile);
    return HW_OK;
}

static int validate_params(const hw_params* p)
{
    if (!p) return fail(HW_E_INVALID, "null params");
    if (!(p->a > 0) || !(p->sigma > 0) || !(p->T_final > 0)) return fail(HW_E_INVALID, "a, sigma, T_final must be > 0");
    if (p->n_steps < 1 || p->n_steps > (1 << 24)) return fail(HW_E_INVALID, "n_steps out of range");
    if (p->n_mat < 2) return fail(HW_E_INVALID, "n_mat < 2");
    if (p->n_steps % (p->n_mat - 1) != 0)                               // common.cuh:25-27
        return fail(HW_E_INVALID, "n_steps must be evenly divisible by (n_mat - 1)");
    if (p->precision != HW_F32 && p->precision != HW_F64) return fail(HW_E_INVALID, "bad precision");
    if (p->table_mode != HW_TABLES_F64 && p->table_mode != HW_TABLES_REF_F32) return fail(HW_E_INVALID, "bad table_mode");
    return HW_OK;
}

int hw_create(const hw_params* p, int device, hw_ctx** out)
{
    if (!out) return fail(HW_E_INVALID, "null out");
    *out = nullptr;
    int rc = validate_params(p);
    if (rc) return rc;
    // hw_comm_* with RCCL exchanges buffers between the devices' contexts: on hosts whose driver only supports dmabuf IPC the
    // legacy mode fails with "hipIpcGetMemHandle: invalid argument".  The HSA runtime reads the variable when it starts, so it
    // is set (never overriding the caller's choice) before the first HIP call a process makes through this library.
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count < 1) return fail(HW_E_HIP, "no HIP device visible");
    if (device < 0) {                                                    // select_gpu, common.cuh:122-141
        size_t best_free = 0;
        device = 0;
        for (int i = 0; i < count; i++) {
            size_t fr = 0, tot = 0;
            if (hipSetDevice(i) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess && fr > best_free) {
                best_free = fr;
                device = i;
            }
        }
    }
    if (device >= count) return fail(HW_E_INVALID, "device %d out of range (%d visible)", device, count);
    HIP_TRY(hipSetDevice(device));
    hw_ctx* c = new (std::nothrow) hw_ctx();
    if (!c) return fail(HW_E_NOMEM, "out of host memory");
    c->p = *p;
    c->device = device;
    hipError_t e = hipGetDeviceProperties(&c->prop, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) c->guard.set_own_stream(c->stream);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&c->d_out, kOutCap * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_out, 2 * kOutCap * sizeof(double), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c->h_out_dev, c->h_out, 0);
    if (e == hipSuccess) {
        c->h_stamps = reinterpret_cast<unsigned long long*>(c->h_out + kOutCap);
        c->h_stamps_dev = reinterpret_cast<unsigned long long*>(c->h_out_dev + kOutCap);
        memset(c->h_stamps, 0, kOutCap * sizeof(unsigned long long));
    }
    if (const char* m = getenv("HWMC_SPIN_WAIT")) c->spin_wait = atoi(m) ? 1 : 0;
    if (e == hipSuccess) e = hipMalloc(&c->d_scale, (size_t)p->n_mat * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&c->d_drift, (size_t)p->n_steps * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->d_clk, hwk::kClkProbes * 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->d_clk, 0, hwk::kClkProbes * 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&c->d_ticket, 9 * hwk::kTicketStride * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(c->d_ticket, 0, 9 * hwk::kTicketStride * sizeof(unsigned));
    if (const char* m = getenv("HWMC_SINGLE_LAUNCH")) c->single_launch = atoi(m) ? 1 : 0;
    if (const char* m = getenv("HWMC_CURVE_GENERIC")) c->curve_generic = atoi(m) ? 1 : 0;
    if (e != hipSuccess) {
        hw_destroy(c);
        return fail(HW_E_HIP, "hw_create: %s", hipGetErrorString(e));
    }
    if (c->prop.warpSize != 64) {
        hw_destroy(c);
        return fail(HW_E_HIP, "device wavefront size is %d; this library is built for wave64 gfx950", c->prop.warpSize);
    }
    *out = c;
    return HW_OK;
}

void hw_destroy(hw_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    // work the context left on another stream: wait on the context's OWN event (recorded when that call ended), never on
    // the caller's stream handle, which may be gone; a pipe's streams are the library's and still alive here
    {
        hipStream_t other = nullptr;
        const auto pend = c->guard.pending(&other);
        if (pend == hw::StreamGuard<hipStream_t>::Pending::TailEvent) (void)hipEventSynchronize(c->ev_tail);
        else if (pend == hw::StreamGuard<hipStream_t>::Pending::OwnedStream) (void)hipStreamSynchronize(other);
        else if (pend == hw::StreamGuard<hipStream_t>::Pending::DeviceSync) (void)hipDeviceSynchronize();
    }
    free_retired(c);
    for (hw_ctx::Stage& sl : c->stage) {
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    }
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->d_scale) (void)hipFree(c->d_scale);
    if (c->d_drift) (void)hipFree(c->d_drift);
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    if (c->d_clk) (void)hipFree(c->d_clk);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t ev : c->prof_ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int hw_get_device_info(hw_ctx* c, hw_device_info* out)
{
    if (!c || !out) return fail(HW_E_INVALID, "null argument");
    memset(out, 0, sizeof *out);
    // (the marketing name comes from libdrm's amdgpu.ids, which some images lack: the architecture string always exists)
    snprintf(out->name, sizeof out->name, "%s (%s)", c->prop.name[0] ? c->prop.name : "AMD GPU", c->prop.gcnArchName);
    out->device = c->device;
    out->compute_units = c->prop.multiProcessorCount;
    out->clock_mhz = c->prop.clockRate / 1000;
    out->wavefront_size = c->prop.warpSize;
    out->lds_per_cu_bytes = (int64_t)c->prop.maxSharedMemoryPerMultiProcessor;
    out->hbm_bytes = (int64_t)c->prop.totalGlobalMem;
    return HW_OK;
}

int hw_get_params(hw_ctx* c, hw_params* out)
{
    if (!c || !out) return fail(HW_E_INVALID, "null argument");
    *out = c->p;
    return HW_OK;
}

int hw_set_market(hw_ctx* c, const float* P, const float* f, int32_t n_mat)
{
    if (!c || !P || !f) return fail(HW_E_INVALID, "null argument");
    if (n_mat != c->p.n_mat) return fail(HW_E_INVALID, "Expected %d floats, got %d", c->p.n_mat, n_mat);
    for (int i = 0; i < n_mat; i++)
        if (!(P[i] > 0.0f) || !std::isfinite(P[i]) || !std::isfinite(f[i]))
            return fail(HW_E_INVALID, "market curve has a non-positive or non-finite entry at %d", i);
    c->P.assign(P, P + n_mat);
    c->f.assign(f, f + n_mat);
    c->scen_cache.clear();
    c->have_market = true;
    return HW_OK;
}

int hw_set_launch(hw_ctx* c, int32_t block_size, int32_t max_blocks_per_cu, int32_t table_in_lds,
                  int32_t pairs_in_flight)
{
    if (!c) return fail(HW_E_INVALID, "null context");
    if (block_size == -1) {
        c->block_size = 0;                                        // back to automatic
    } else if (block_size) {
        if (block_size < 64 || block_size > 1024 || block_size % 64) return fail(HW_E_INVALID, "block_size must be a multiple of 64 in [64,1024]");
        c->block_size = block_size;
    }
    if (max_blocks_per_cu < 0) return fail(HW_E_INVALID, "max_blocks_per_cu < 0");
    if (pairs_in_flight < -1 || pairs_in_flight > 2)
        return fail(HW_E_INVALID, "pairs_in_flight must be 1 or 2 (0 keeps the current choice, -1 returns to automatic)");
    c->max_blocks_per_cu = max_blocks_per_cu;
    if (pairs_in_flight > 0) c->ilp = pairs_in_flight;
    else if (pairs_in_flight == -1) c->ilp = 0;
    c->table_in_lds = table_in_lds ? 1 : 0;
    return HW_OK;
}

// ---- ZBC ----------------------------------------------------------------------------------
int hw_zbc_moments(hw_ctx* c, const hw_zbc_request* r, double out[5], float* kernel_ms)
{
    if (!c || !r || !out) return fail(HW_E_INVALID, "null argument");
    return timed_sync(c, 5, out, kernel_ms, [&](double* d_out, hipStream_t st) {
        return zbc_enqueue(c, r, 1, nullptr, nullptr, d_out, st);
    });
}

int hw_zbc_moments_async(hw_ctx* c, const hw_zbc_request* r, double* d_out, void* stream)
{
    if (!c || !r || !d_out) return fail(HW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return zbc_enqueue(c, r, 1, nullptr, nullptr, d_out, stream ? (hipStream_t)stream : c->stream);
}

int hw_zbc_scenarios(hw_ctx* c, const hw_zbc_request* r, int32_t n_scen, const double* sigma,
                     const int32_t* drift_mode, double* out, float* kernel_ms)
{
    if (!c || !r || !out || !sigma || !drift_mode) return fail(HW_E_INVALID, "null argument");
    if (n_scen < 1 || n_scen > hwk::kMaxScen) return fail(HW_E_INVALID, "n_scen must be 1..%d", hwk::kMaxScen);
    return timed_sync(c, 5 * n_scen, out, kernel_ms, [&](double* d_out, hipStream_t st) {
        return zbc_enqueue(c, r, n_scen, sigma, drift_mode, d_out, st);
    });
}

// n_runs independent seeds of one request in ONE launch, enqueue only (shared by the single-context and the fleet entry points)
static int zbc_runs_enqueue(hw_ctx* c, const hw_zbc_request* r, int32_t n_runs, uint64_t seed_stride, double* d_out, hipStream_t st)
{
    if (n_runs < 1 || 5 * n_runs > kOutCap) return fail(HW_E_INVALID, "n_runs must be 1..%d", kOutCap / 5);
    if (r->stepper != HW_STEPPER_SUPERPOSED) return fail(HW_E_INVALID, "batched runs use the SUPERPOSED stepper");
    int n_steps_S1 = 0;
    int rc = check_request(c, r, &n_steps_S1);
    if (rc) return rc;
    Scenario sc;
    rc = build_scenario(c, r->sigma, r->drift_mode, n_steps_S1, r->S1, r->S2, &sc);
    if (rc) return rc;
    return on_stream(c, st, [&] {
        return c->p.precision == HW_F64 ? launch_zbc_superposed<double, 1>(c, r, n_steps_S1, &sc, d_out, st, n_runs, seed_stride)
                                        : launch_zbc_superposed<float, 1>(c, r, n_steps_S1, &sc, d_out, st, n_runs, seed_stride);
    });
}

int hw_zbc_moments_runs(hw_ctx* c, const hw_zbc_request* r, int32_t n_runs, uint64_t seed_stride, double* out,
                        float* kernel_ms)
{
    if (!c || !r || !out) return fail(HW_E_INVALID, "null argument");
    if (n_runs < 1 || 5 * n_runs > kOutCap) return fail(HW_E_INVALID, "n_runs must be 1..%d", kOutCap / 5);
    return timed_sync(c, 5 * n_runs, out, kernel_ms, [&](double* d_out, hipStream_t st) {
        return zbc_runs_enqueue(c, r, n_runs, seed_stride, d_out, st);
    });
}

// ---- curve --------------------------------------------------------------------------------
static int zcb_enqueue(hw_ctx* c, int64_t n_pairs, uint64_t seed, uint64_t path_offset, double sigma,
                       double* d_P_sum, hipStream_t st)
{
    if (n_pairs < 0) return fail(HW_E_INVALID, "n_pairs < 0");
    return on_stream(c, st, [&] {
        return c->p.precision == HW_F64 ? launch_zcb<double>(c, n_pairs, seed, path_offset, sigma, d_P_sum, st)
                                        : launch_zcb<float>(c, n_pairs, seed, path_offset, sigma, d_P_sum, st);
    });
}

int hw_zcb_curve(hw_ctx* c, int64_t n_pairs, uint64_t seed, uint64_t path_offset, double sigma,
                 double* P_sum, float* kernel_ms)
{
    if (!c || !P_sum) return fail(HW_E_INVALID, "null argument");
    return timed_sync(c, c->p.n_mat, P_sum, kernel_ms, [&](double* d_out, hipStream_t st) {
        return zcb_enqueue(c, n_pairs, seed, path_offset, sigma, d_out, st);
    });
}

int hw_zcb_curve_async(hw_ctx* c, int64_t n_pairs, uint64_t seed, uint64_t path_offset, double sigma,
                       double* d_P_sum, void* stream)
{
    if (!c || !d_P_sum) return fail(HW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return zcb_enqueue(c, n_pairs, seed, path_offset, sigma, d_P_sum, stream ? (hipStream_t)stream : c->stream);
}

// ---- Vega ---------------------------------------------------------------------------------
static int vega_enqueue(hw_ctx* c, int64_t n_paths, uint64_t seed, uint64_t path_offset, double S1,
                        double S2, double K, int antithetic, double* d_out, hipStream_t st)
{
    hw_zbc_request r{};
    r.n_pairs = n_paths; r.S1 = S1; r.S2 = S2; r.K = K;
    int n_steps_S1 = 0;
    int rc = check_request(c, &r, &n_steps_S1);
    if (rc) return rc;
    return on_stream(c, st, [&] {
        return c->p.precision == HW_F64
                   ? launch_vega<double>(c, n_paths, seed, path_offset, S1, S2, K, antithetic, n_steps_S1, d_out, st)
                   : launch_vega<float>(c, n_paths, seed, path_offset, S1, S2, K, antithetic, n_steps_S1, d_out, st);
    });
}

int hw_vega_pathwise(hw_ctx* c, int64_t n_paths, uint64_t seed, uint64_t path_offset, double S1,
                     double S2, double K, int32_t antithetic, double out[1], float* kernel_ms)
{
    if (!c || !out) return fail(HW_E_INVALID, "null argument");
    return timed_sync(c, 1, out, kernel_ms, [&](double* d_out, hipStream_t st) {
        return vega_enqueue(c, n_paths, seed, path_offset, S1, S2, K, antithetic, d_out, st);
    });
}

int hw_vega_pathwise_runs(hw_ctx* c, int64_t n_paths, uint64_t seed, uint64_t path_offset, double S1, double S2,
                          double K, int32_t antithetic, int32_t n_runs, uint64_t seed_stride, double* out, float* kernel_ms)
{
    if (!c || !out) return fail(HW_E_INVALID, "null argument");
    if (n_runs < 1 || n_runs > kOutCap) return fail(HW_E_INVALID, "n_runs must be 1..%d", kOutCap);
    hw_zbc_request r{};
    r.n_pairs = n_paths; r.S1 = S1; r.S2 = S2; r.K = K;
    int n_steps_S1 = 0;
    int rc = check_request(c, &r, &n_steps_S1);
    if (rc) return rc;
    return timed_sync(c, n_runs, out, kernel_ms, [&](double* d_out, hipStream_t st) {
        return on_stream(c, st, [&] {
            return c->p.precision == HW_F64
                       ? launch_vega<double>(c, n_paths, seed, path_offset, S1, S2, K, antithetic, n_steps_S1, d_out, st, n_runs, seed_stride)
                       : launch_vega<float>(c, n_paths, seed, path_offset, S1, S2, K, antithetic, n_steps_S1, d_out, st, n_runs, seed_stride);
        });
    });
}

int hw_vega_pathwise_async(hw_ctx* c, int64_t n_paths, uint64_t seed, uint64_t path_offset, double S1,
                           double S2, double K, int32_t antithetic, double* d_out, void* stream)
{
    if (!c || !d_out) return fail(HW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return vega_enqueue(c, n_paths, seed, path_offset, S1, S2, K, antithetic, d_out,
                        stream ? (hipStream_t)stream : c->stream);
}

// ---- small kernels ------------------------------------------------------------------------
int hw_sample_paths(hw_ctx* c, int32_t n, uint64_t seed, uint64_t path_offset, float* out)
{
    if (!c || !out) return fail(HW_E_INVALID, "null argument");
    if (n < 0 || n > (1 << 20)) return fail(HW_E_INVALID, "n out of range");
    if (n == 0) return HW_OK;
    HIP_TRY(hipSetDevice(c->device));
    const hw_params& p = c->p;
    const hw::StepConstants k = hw::step_constants(p, p.sigma);
    std::vector<double> drift;
    hw::drift_tables(p, p.sigma, &drift, nullptr);
    std::vector<float> tab(p.n_steps);
    for (int i = 0; i < p.n_steps; i++) tab[i] = (float)drift[i];
    const size_t bytes = (size_t)n * (p.n_steps + 1) * sizeof(float);
    float* d_paths = nullptr;
    HIP_TRY(hipMalloc(&d_paths, bytes));
    c->drift_valid = false;                                               // d_drift now holds the full base table
    hipError_t e = hipSuccess;
    if (use_stream(c, c->stream) != HW_OK || stage_upload(c, c->d_drift, tab.data(), tab.size() * sizeof(float), c->stream) != HW_OK) {
        (void)hipFree(d_paths);
        return HW_E_HIP;
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hwk::sample_paths_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, d_paths,
                           (const float*)c->d_drift, n, p.n_steps, seed, path_offset, (float)p.r0, (float)k.e,
                           (float)k.sig_st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_paths, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_paths);
    if (e != hipSuccess) return fail(HW_E_HIP, "hw_sample_paths: %s", hipGetErrorString(e));
    return HW_OK;
}

int hw_debug_normals(hw_ctx* c, uint64_t seed, uint64_t subseq, int32_t n_blocks, float* out)
{
    if (!c || !out) return fail(HW_E_INVALID, "null argument");
    if (n_blocks < 1 || n_blocks > (1 << 20)) return fail(HW_E_INVALID, "n_blocks out of range");
    HIP_TRY(hipSetDevice(c->device));
    float* d_n = nullptr;
    uint32_t* d_raw = nullptr;
    HIP_TRY(hipMalloc(&d_n, (size_t)n_blocks * 4 * sizeof(float)));
    hipError_t e = hipMalloc(&d_raw, (size_t)n_blocks * 8 * sizeof(uint32_t));
    std::vector<uint32_t> raw((size_t)n_blocks * 8);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hwk::debug_normals_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, c->stream, d_n, d_raw,
                           seed, subseq, n_blocks);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_n, (size_t)n_blocks * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(raw.data(), d_raw, raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_n);
    if (d_raw) (void)hipFree(d_raw);
    if (e != hipSuccess) return fail(HW_E_HIP, "hw_debug_normals: %s", hipGetErrorString(e));
    for (int j = 0; j < n_blocks; j++)
        for (int w = 0; w < 4; w++)
            if (raw[8 * j + w] != raw[8 * j + 4 + w])
                return fail(HW_E_HIP, "device Philox self-check failed: hoisted and plain rounds differ at block %d", j);
    return HW_OK;
}

int hw_debug_raw(hw_ctx* c, uint64_t seed, uint64_t subseq, int32_t n_blocks, uint32_t* out)
{
    if (!c || !out) return fail(HW_E_INVALID, "null argument");
    if (n_blocks < 1 || n_blocks > (1 << 20)) return fail(HW_E_INVALID, "n_blocks out of range");
    HIP_TRY(hipSetDevice(c->device));
    float* d_n = nullptr;
    uint32_t* d_raw = nullptr;
    HIP_TRY(hipMalloc(&d_n, (size_t)n_blocks * 4 * sizeof(float)));
    hipError_t e = hipMalloc(&d_raw, (size_t)n_blocks * 8 * sizeof(uint32_t));
    std::vector<uint32_t> raw((size_t)n_blocks * 8);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hwk::debug_normals_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, c->stream, d_n, d_raw,
                           seed, subseq, n_blocks);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(raw.data(), d_raw, raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_n);
    if (d_raw) (void)hipFree(d_raw);
    if (e != hipSuccess) return fail(HW_E_HIP, "hw_debug_raw: %s", hipGetErrorString(e));
    for (int j = 0; j < n_blocks; j++)
        for (int w = 0; w < 4; w++) out[4 * j + w] = raw[8 * j + w];
    return HW_OK;
}

int hw_debug_exp2(hw_ctx* c, const double* x, int32_t n, double* out)
{
    if (!c || !x || !out) return fail(HW_E_INVALID, "null argument");
    if (n < 1 || n > (1 << 24)) return fail(HW_E_INVALID, "n out of range");
    HIP_TRY(hipSetDevice(c->device));
    double *d_x = nullptr, *d_o = nullptr;
    HIP_TRY(hipMalloc(&d_x, (size_t)n * sizeof(double)));
    hipError_t e = hipMalloc(&d_o, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_x, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hwk::debug_exp2_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const double*)d_x, d_o, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_x);
    if (d_o) (void)hipFree(d_o);
    if (e != hipSuccess) return fail(HW_E_HIP, "hw_debug_exp2: %s", hipGetErrorString(e));
    return HW_OK;
}

int hw_profile_enable(hw_ctx* c, int32_t max_launches)
{
    if (!c) return fail(HW_E_INVALID, "null context");
    if (max_launches < 0 || max_launches > 100000) return fail(HW_E_INVALID, "max_launches out of range");
    HIP_TRY(hipSetDevice(c->device));
    while ((int)c->prof_ev.size() < 2 * max_launches) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        c->prof_ev.push_back(ev);
    }
    c->prof_used = 0;
    c->prof_cap = max_launches;
    return HW_OK;
}

int hw_profile_read(hw_ctx* c, float* ms, int32_t cap, int32_t* n_out)
{
    if (!c || !ms || !n_out) return fail(HW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const int n = std::min(c->prof_used, (int)cap);
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipEventSynchronize(c->prof_ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms[i], c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
    }
    *n_out = n;
    c->prof_cap = 0;
    c->prof_used = 0;
    return HW_OK;
}

int hw_profile_spans(hw_ctx* c, hw_ctx* ref, float* start_ms, float* end_ms, int32_t cap, int32_t* n_out)
{
    if (!c || !ref || !start_ms || !end_ms || !n_out) return fail(HW_E_INVALID, "null argument");
    if (c->device != ref->device) return fail(HW_E_INVALID, "contexts on different devices have no common event clock");
    if (ref->prof_used < 1) return fail(HW_E_INVALID, "the reference context has no profiled launch");
    HIP_TRY(hipSetDevice(c->device));
    const int n = std::min(c->prof_used, (int)cap);
    HIP_TRY(hipEventSynchronize(ref->prof_ev[0]));
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipEventSynchronize(c->prof_ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&start_ms[i], ref->prof_ev[0], c->prof_ev[2 * i]));
        HIP_TRY(hipEventElapsedTime(&end_ms[i], ref->prof_ev[0], c->prof_ev[2 * i + 1]));
    }
    *n_out = n;
    return HW_OK;
}

int hw_get_launch_geometry(hw_ctx* c, const char* kernel, int64_t n_units, int32_t* block, int32_t* grid, int32_t* iters,
                           int64_t* partial_bytes)
{
    if (!c || !kernel) return fail(HW_E_INVALID, "null argument");
    if (n_units < 0) return fail(HW_E_INVALID, "n_units < 0");
    Geometry g{};
    int64_t row = 0;
    // (the automatic choice of pairs in flight is evaluated for the reference option: S1 = T_final / 2)
    const int n_half = std::max(1, hw::steps_to(c->p, c->p.T_final / 2.0));
    const int ilp_auto = effective_ilp(c, n_units, n_half, hw::step_constants(c->p, c->p.sigma).e, 1);
    if (!strcmp(kernel, "zbc") || !strcmp(kernel, "zbc_ref")) { g = geometry(c, n_units, !strcmp(kernel, "zbc") ? ilp_auto : 1); row = 5 * 8; }
    else if (!strcmp(kernel, "vega")) { g = geometry(c, n_units, effective_ilp(c, n_units, n_half, hw::step_constants(c->p, c->p.sigma).e, 1, 4)); row = 8; }
    else if (!strcmp(kernel, "zcb") || !strcmp(kernel, "zcb_blocked")) {
        int P = 1;
        if (!strcmp(kernel, "zcb_blocked")) {
            const int64_t resident = (int64_t)c->prop.multiProcessorCount * 2048;
            P = n_units >= 8 * resident ? 2 : 1;
        }
        g = geometry(c, n_units, P, true);
        row = (int64_t)c->p.n_mat * 8;
    } else return fail(HW_E_INVALID, "unknown kernel '%s'", kernel);
    if (block) *block = g.block;
    if (grid) *grid = g.grid;
    if (iters) *iters = g.iters;
    if (partial_bytes) *partial_bytes = row * g.grid;
    return HW_OK;
}

int hw_profile_clock_samples(hw_ctx* c, double* mhz, double* start_us, double* end_us, int32_t cap, int32_t* n_out)
{
    if (!c || !n_out) return fail(HW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[hwk::kClkProbes * 4];
    HIP_TRY(hipMemcpy(h, c->d_clk, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int j = 0; j < hwk::kClkProbes; j++)
        if (h[4 * j + 3] > h[4 * j + 1]) t0 = std::min(t0, h[4 * j + 1]);
    int n = 0;
    for (int j = 0; j < hwk::kClkProbes && n < cap; j++) {
        if (h[4 * j + 3] <= h[4 * j + 1]) continue;               // this workgroup did not probe (grid smaller than kClkProbes)
        // s_memtime ticks at the shader clock, s_memrealtime at a constant 100 MHz
        if (mhz) mhz[n] = (double)(h[4 * j + 2] - h[4 * j]) / (double)(h[4 * j + 3] - h[4 * j + 1]) * 100.0;
        if (start_us) start_us[n] = (double)(h[4 * j + 1] - t0) / 100.0;
        if (end_us) end_us[n] = (double)(h[4 * j + 3] - t0) / 100.0;
        n++;
    }
    *n_out = n;
    return HW_OK;
}

int hw_profile_clock(hw_ctx* c, double* shader_mhz)
{
    if (!c || !shader_mhz) return fail(HW_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[hwk::kClkProbes * 4];
    HIP_TRY(hipMemcpy(h, c->d_clk, sizeof h, hipMemcpyDeviceToHost));
    // time-weighted mean over the probing workgroups: they are spread over the grid, i.e. along the launch
    double cyc = 0.0, real = 0.0;
    for (int j = 0; j < hwk::kClkProbes; j++)
        if (h[4 * j + 3] > h[4 * j + 1]) {
            cyc += (double)(h[4 * j + 2] - h[4 * j]);
            real += (double)(h[4 * j + 3] - h[4 * j + 1]);
        }
    if (!(real > 0.0)) return fail(HW_E_INVALID, "no clock probe recorded (profile a superposed ZBC or a blocked curve launch first)");
    *shader_mhz = cyc / real * 100.0;
    return HW_OK;
}

int hw_get_kernel_info(hw_ctx* c, const char* kernel, hw_kernel_info* out)
{
    if (!c || !kernel || !out) return fail(HW_E_INVALID, "null argument");
    const void* fn = nullptr;
    const bool f64 = c->p.precision == HW_F64;
    if (!strcmp(kernel, "zbc"))
        fn = f64 ? (const void*)hwk::zbc_superposed_kernel<double, 1, 1> : (const void*)hwk::zbc_superposed_kernel<float, 1, 1>;
    else if (!strcmp(kernel, "zbc_ref"))
        fn = c->table_in_lds ? (const void*)hwk::zbc_reference_kernel<1> : (const void*)hwk::zbc_reference_kernel<0>;
    else if (!strcmp(kernel, "zcb"))
        fn = f64 ? (const void*)hwk::zcb_curve_kernel<double> : (const void*)hwk::zcb_curve_kernel<float>;
    else if (!strcmp(kernel, "zcb_blocked"))
        fn = f64 ? (const void*)hwk::zcb_curve_blocked_kernel<1, double> : (const void*)hwk::zcb_curve_blocked_kernel<1, float>;
    else if (!strcmp(kernel, "vega"))
        fn = f64 ? (const void*)hwk::vega_pathwise_kernel<double, 1> : (const void*)hwk::vega_pathwise_kernel<float, 1>;
    else
        return fail(HW_E_INVALID, "unknown kernel '%s'", kernel);
    HIP_TRY(hipSetDevice(c->device));
    hipFuncAttributes attr{};
    HIP_TRY(hipFuncGetAttributes(&attr, fn));
    memset(out, 0, sizeof *out);
    out->vgprs = attr.numRegs;
    out->sgprs = 0;
    out->lds_bytes = (int32_t)attr.sharedSizeBytes;
    out->scratch_bytes = (int32_t)attr.localSizeBytes;
    out->max_threads_per_block = attr.maxThreadsPerBlock;
    out->block_size = c->block_size ? c->block_size : 256;
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, out->block_size, 0));
    out->blocks_per_cu = nb;
    return HW_OK;
}

// ---- host-side pieces ----------------------------------------------------------------------
int hw_drift_tables(hw_ctx* c, double sigma, int32_t drift_mode, double* drift, double* sigma_drift)
{
    if (!c) return fail(HW_E_INVALID, "null context");
    if (!(sigma > 0.0)) sigma = c->p.sigma;
    std::vector<double> d, sd;
    if (drift_mode == HW_DRIFT_SHIFTED) {
        hw::shifted_drift_table(c->p, sigma, c->p.sigma, &d);
        hw::drift_tables(c->p, sigma, nullptr, &sd);
    } else if (drift_mode == HW_DRIFT_BASE) {
        hw::drift_tables(c->p, sigma, &d, &sd);
    } else {
        return fail(HW_E_INVALID, "bad drift_mode");
    }
    if (drift) std::copy(d.begin(), d.end(), drift);
    if (sigma_drift) std::copy(sd.begin(), sd.end(), sigma_drift);
    return HW_OK;
}

int hw_average_and_forward(const hw_params* p, const double* P_sum, int64_t n_total, float* P, float* f)
{
    int rc = validate_params(p);
    if (rc) return rc;
    if (!P_sum || !P || !f || n_total < 1) return fail(HW_E_INVALID, "bad argument");
    {
#pragma clang fp contract(off)
    const int n_mat = p->n_mat;
    const float inv_dT = 1 / ((float)p->T_final / (float)(n_mat - 1));              // 1_bond:76
    for (int j = 0; j < n_mat; j++) P[j] = (float)(P_sum[j] / (double)n_total);     // market_data.cuh:114
    for (int j = 0; j < n_mat; j++) {
        const int lo = (j == 0) ? 0 : j - 1;                                        // market_data.cuh:122-125
        const int hi = (j == n_mat - 1) ? n_mat - 1 : j + 1;
        const float scale = (j == 0 || j == n_mat - 1) ? 1.0f : 0.5f;
        f[j] = -scale * inv_dT * (logf(P[hi]) - logf(P[lo]));
    }
    }
    return HW_OK;
}

int hw_recover_theta(const hw_params* p, const float* f, float* theta_rec, float* theta_orig, float* Ts)
{
    int rc = validate_params(p);
    if (rc) return rc;
    if (!f || !theta_rec || !theta_orig || !Ts) return fail(HW_E_INVALID, "null argument");
    {
#pragma clang fp contract(off)
    const int n = p->n_mat;
    const float a = (float)p->a, sg = (float)p->sigma;
    const float spacing = (float)p->T_final / (float)(n - 1);
    for (int i = 0; i < n; i++) {
        const float T = (float)i * spacing;                                         // 2_option:25
        float d;
        if (i == 0) d = (f[1] - f[0]) / spacing;                                    // common.cuh:250-258
        else if (i == n - 1) d = (f[i] - f[i - 1]) / spacing;
        else d = (f[i + 1] - f[i - 1]) / (2.0f * spacing);
        const float convexity = (sg * sg / (2.0f * a)) * (1.0f - expf(-2.0f * a * T)); // 2_option:28-29
        theta_rec[i] = d + a * f[i] + convexity;                                    // 2_option:31
        theta_orig[i] = (T < (float)p->th_break) ? ((float)p->th_a1 + (float)p->th_b1 * T)
                                                 : ((float)p->th_a2 + (float)p->th_b2 * T);
        Ts[i] = T;
    }
    }
    return HW_OK;
}

int hw_cv_price(const double sums[5], int64_t n_total, double P0S2, double res[9])
{
    if (!sums || !res || n_total < 1) return fail(HW_E_INVALID, "bad argument");
    const double N = (double)n_total;
    const double mX = sums[0] / N, mY = sums[1] / N;                                // 2_option:157-158
    const double varY = sums[3] / N - mY * mY;                                      // 2_option:161-163
    const double cov = sums[4] / N - mX * mY;                                       // 2_option:166-168
    const double beta = cov / varY;                                                 // 2_option:171
    const double adj = beta * (mY - P0S2);                                          // 2_option:174
    const double varX = sums[2] / N - mX * mX;                                      // 2_option:277-278
    res[0] = mX; res[1] = mY; res[2] = varY; res[3] = cov; res[4] = beta; res[5] = adj;
    res[6] = mX - adj;                                                              // 2_option:175
    res[7] = varX;
    res[8] = cov / std::sqrt(varX * varY);                                          // 2_option:281
    return HW_OK;
}

int hw_analytic_curve(const hw_params* p, float* P, float* f)
{
    int rc = validate_params(p);
    if (rc) return rc;
    if (!P || !f) return fail(HW_E_INVALID, "null argument");
    const double spacing = p->T_final / (p->n_mat - 1);
    for (int j = 0; j < p->n_mat; j++) {
        double Pv, fv;
        hw::analytic_curve(*p, j * spacing, &Pv, &fv);
        P[j] = (float)Pv;
        f[j] = (float)fv;
    }
    return HW_OK;
}


}  // extern "C"

// ---- a stream of independent pricing calls, several hardware queues in flight (hw_pipe_*) ---------------------
struct hw_pipe {
    int device = 0;
    std::vector<hw_ctx*> ctx;            // one per queue: a context's workspaces belong to one launch at a time
    std::vector<hipStream_t> cand;       // candidate streams; chosen[q] indexes the one queue q uses
    std::vector<int> chosen;
    std::vector<hipEvent_t> ev;          // per queue: "everything enqueued on this queue so far"
    double* h_out = nullptr;             // result rows in pinned, device-visible host memory: every column pass stores its five
    double* h_out_dev = nullptr;         // sums straight to the host, no copy behind the batch
    size_t out_cap = 0;                  // doubles
    bool tuned = false;
};

namespace {

constexpr int kPipeCandidates = 8;

int pipe_ensure_out(hw_pipe* pp, size_t doubles)
{
    if (doubles <= pp->out_cap) return HW_OK;
    HIP_TRY(hipDeviceSynchronize());                       // (only ever grows between calls: nothing is in flight)
    if (pp->h_out) HIP_TRY(hipHostFree(pp->h_out));
    pp->h_out = pp->h_out_dev = nullptr;
    pp->out_cap = 0;
    HIP_TRY(hipHostMalloc((void**)&pp->h_out, doubles * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&pp->h_out_dev, pp->h_out, 0));
    pp->out_cap = doubles;
    return HW_OK;
}

// n requests alternating over the queues `use` (indices into cand), results in d_out rows; waits for all of them
int pipe_run(hw_pipe* pp, const std::vector<int>& use, const hw_zbc_request* reqs, int n, bool same_request, double* host_out)
{
    const int Q = (int)use.size();
    for (int i = 0; i < n; i++) {
        const int q = i % Q;
        const int rc = zbc_enqueue(pp->ctx[q], same_request ? reqs : reqs + i, 1, nullptr, nullptr, pp->h_out_dev + (size_t)5 * i, pp->cand[use[q]]);
        if (rc) {
            (void)hipDeviceSynchronize();                  // leave nothing in flight behind an error
            return rc;
        }
    }
    hipStream_t s0 = pp->cand[use[0]];
    hipError_t e = hipSuccess;
    for (int q = 1; q < Q && q < n && e == hipSuccess; q++) {   // queue 0 waits for the others: ONE host synchronisation
        e = hipEventRecord(pp->ev[q], pp->cand[use[q]]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s0, pp->ev[q], 0);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s0);
    if (e != hipSuccess) {
        (void)hipDeviceSynchronize();                      // the header's promise: nothing in flight behind an error
        return fail(HW_E_HIP, "hw_pipe: %s", hipGetErrorString(e));
    }
    if (host_out && n > 0) memcpy(host_out, pp->h_out, (size_t)5 * n * sizeof(double));
    for (hw_ctx* c : pp->ctx) free_retired(c);
    return HW_OK;
}

}  // namespace

extern "C" {

int hw_pipe_create(const hw_params* p, int device, int32_t n_queues, hw_pipe** out)
{
    if (!out) return fail(HW_E_INVALID, "null out");
    *out = nullptr;
    if (n_queues < 1 || n_queues > 4) return fail(HW_E_INVALID, "n_queues must be 1..4");
    hw_pipe* pp = new (std::nothrow) hw_pipe();
    if (!pp) return fail(HW_E_NOMEM, "out of host memory");
    for (int q = 0; q < n_queues; q++) {
        hw_ctx* c = nullptr;
        const int rc = hw_create(p, q == 0 ? device : pp->device, &c);
        if (rc) { hw_pipe_destroy(pp); return rc; }
        c->guard.set_all_streams_owned(true);             // the pipe's candidate streams are destroyed after its contexts
        pp->ctx.push_back(c);
        pp->device = c->device;                            // (device < 0: every queue on the device the first context picked)
    }
    hipError_t e = hipSetDevice(pp->device);
    for (int k = 0; k < (n_queues > 1 ? kPipeCandidates : 1) && e == hipSuccess; k++) {
        hipStream_t st = nullptr;
        e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) pp->cand.push_back(st);
    }
    for (int q = 0; q < n_queues && e == hipSuccess; q++) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) pp->ev.push_back(ev);
        pp->chosen.push_back(q);
    }
    if (e != hipSuccess) {
        hw_pipe_destroy(pp);
        return fail(HW_E_HIP, "hw_pipe_create: %s", hipGetErrorString(e));
    }
    pp->tuned = n_queues == 1;
    *out = pp;
    return HW_OK;
}

void hw_pipe_destroy(hw_pipe* pp)
{
    if (!pp) return;
    (void)hipSetDevice(pp->device);
    (void)hipDeviceSynchronize();
    for (hw_ctx* c : pp->ctx) hw_destroy(c);               // (before the streams: a context may still name one as its last)
    for (hipEvent_t ev : pp->ev) (void)hipEventDestroy(ev);
    for (hipStream_t st : pp->cand) (void)hipStreamDestroy(st);
    if (pp->h_out) (void)hipHostFree(pp->h_out);
    delete pp;
}

int hw_pipe_set_market(hw_pipe* pp, const float* P, const float* f, int32_t n_mat)
{
    if (!pp) return fail(HW_E_INVALID, "null pipe");
    for (hw_ctx* c : pp->ctx) {
        const int rc = hw_set_market(c, P, f, n_mat);
        if (rc) return rc;
    }
    return HW_OK;
}

int hw_pipe_queue(hw_pipe* pp, int32_t q, hw_ctx** ctx_out, int32_t* stream_index_out, int32_t* n_queues_out)
{
    if (!pp) return fail(HW_E_INVALID, "null pipe");
    if (q < 0 || q >= (int32_t)pp->ctx.size()) return fail(HW_E_INVALID, "queue %d out of range (the pipe has %zu)", q, pp->ctx.size());
    if (ctx_out) *ctx_out = pp->ctx[q];
    if (stream_index_out) *stream_index_out = pp->chosen[q];
    if (n_queues_out) *n_queues_out = (int32_t)pp->ctx.size();
    return HW_OK;
}

int hw_pipe_tune(hw_pipe* pp, const hw_zbc_request* sample, int32_t launches, float* best_ms_per_call)
{
    if (!pp || !sample) return fail(HW_E_INVALID, "null argument");
    int n_steps_S1 = 0;
    for (hw_ctx* c : pp->ctx) {
        const int rc = check_request(c, sample, &n_steps_S1);
        if (rc) return rc;
    }
    if (launches <= 0) launches = 16;
    HIP_TRY(hipSetDevice(pp->device));
    int rc = pipe_ensure_out(pp, (size_t)5 * launches);
    if (rc) return rc;
    const int Q = (int)pp->ctx.size(), C = (int)pp->cand.size();
    auto trial = [&](const std::vector<int>& use, float* ms) -> int {
        float best = 1e30f;
        for (int rep = 0; rep < 2; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            const int r = pipe_run(pp, use, sample, launches, true, nullptr);
            if (r) return r;
            best = std::min(best, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / launches);
        }
        *ms = best;
        return HW_OK;
    };
    {   // bring the clocks up on the caller's workload before anything is compared (~100 ms)
        const auto t0 = std::chrono::steady_clock::now();
        float ms = 0.0f;
        while (std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() < 100.0f) {
            rc = trial(std::vector<int>(1, pp->chosen[0]), &ms);
            if (rc) return rc;
        }
    }
    std::vector<int> use;
    float best_ms = 1e30f;
    if (Q == 1) {
        use.assign(1, pp->chosen[0]);
        rc = trial(use, &best_ms);
        if (rc) return rc;
    } else {
        // every pair of candidates, then greedily one more stream per further queue
        for (int i = 0; i < C; i++)
            for (int j = i + 1; j < C; j++) {
                float ms = 0.0f;
                rc = trial({i, j}, &ms);
                if (rc) return rc;
                if (ms < best_ms) { best_ms = ms; use = {i, j}; }
            }
        while ((int)use.size() < Q) {
            int pick = -1;
            float pick_ms = 1e30f;
            for (int k = 0; k < C; k++) {
                if (std::find(use.begin(), use.end(), k) != use.end()) continue;
                std::vector<int> tryu = use;
                tryu.push_back(k);
                float ms = 0.0f;
                rc = trial(tryu, &ms);
                if (rc) return rc;
                if (ms < pick_ms) { pick_ms = ms; pick = k; }
            }
            use.push_back(pick);
            best_ms = pick_ms;
        }
    }
    pp->chosen = use;
    pp->tuned = true;
    if (best_ms_per_call) *best_ms_per_call = best_ms;
    return HW_OK;
}

int hw_pipe_zbc_moments(hw_pipe* pp, const hw_zbc_request* reqs, int32_t n, double* out, float* wall_ms)
{
    if (!pp || (n > 0 && (!reqs || !out))) return fail(HW_E_INVALID, "null argument");
    if (n < 0) return fail(HW_E_INVALID, "n < 0");
    if (wall_ms) *wall_ms = 0.0f;
    if (n == 0) return HW_OK;
    for (int i = 0; i < n; i++) {                          // refuse the whole batch before anything is enqueued
        int n_steps_S1 = 0;
        const int rc = check_request(pp->ctx[i % pp->ctx.size()], reqs + i, &n_steps_S1);
        if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(pp->device));
    if (!pp->tuned) {
        // automatic tuning on the first request: which streams share a hardware queue does not depend on the batch size,
        // so the sample is capped at 2^20 pairs and 8 launches per trial (~0.1 s whatever the caller's request;
        // hw_pipe_tune called by hand uses the request as given)
        hw_zbc_request sample = reqs[0];
        sample.n_pairs = std::min<int64_t>(sample.n_pairs, (int64_t)1 << 20);
        const int rc = hw_pipe_tune(pp, &sample, 8, nullptr);
        if (rc) return rc;
    }
    int rc = pipe_ensure_out(pp, (size_t)5 * n);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    rc = pipe_run(pp, pp->chosen, reqs, n, false, out);
    if (wall_ms) *wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

}  // extern "C"

// ---- multi-GPU in one process: shard over several contexts, one RCCL all-reduce over xGMI ----------
// One PERSISTENT host thread per device (created by hw_comm_create, bound to its device once): a pricing call hands
// every worker the same job; worker g enqueues its shard's kernels on context g's stream, enqueues the all-reduce on
// the same stream, and waits for that stream.  The devices therefore start within the wake-up jitter of the workers
// (they spin for a short while after each job before they sleep on a condition variable) instead of one launch
// latency after another: the single-threaded version of round 1 walked the devices in a loop, ~5 us of skew per
// device, which alone capped the strong scaling of a 2^24-pair call at ~6.6x on 8 GPUs.
struct hw_comm {
    std::vector<hw_ctx*> ctx;
    std::vector<ncclComm_t> comm;
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // G persistent worker threads + the two-phase job protocol (rendezvous between a device's own work and the collective,
    // abort for failures inside the collective): hw_fleet.hpp, checked on the CPU under ThreadSanitizer with a mock collective
    hw::Fleet fleet;
    // HW_COMM_HOST_SUM: no RCCL; every worker brings its shard's sums to its slot here and the caller adds the slots
    // in rank order (deterministic).  Contexts may then share a device: the threaded dispatch can be rehearsed, G > 1,
    // on a box with one GPU.
    bool host_sum = false;
    std::vector<double> host_slots;        // [G][kOutCap]
    // failure injection for the rehearsal tests: HWMC_COMM_INJECT_FAIL=rank:phase[:count], read by hw_comm_create*
    int inject_rank = -1, inject_phase = 0;
    std::atomic<int> inject_left{0};
};

// shard [0, n) in contiguous blocks: rank g gets {offset, count}
static void comm_shard(int64_t n, int g, int G, int64_t* off, int64_t* cnt)
{
    const int64_t base = n / G, rem = n % G;
    *off = g * base + std::min<int64_t>(g, rem);
    *cnt = base + (g < rem ? 1 : 0);
}

// run job on every worker, wait for all of them; the first failing rank's code and text reach the caller
static int comm_dispatch(hw_comm* cm, const hw::Fleet::Job& job)
{
    int rc = HW_OK;
    std::string text;
    const int g = cm->fleet.dispatch(job, &rc, &text);
    if (g < 0) return HW_OK;
    return fail(rc, "device %d: %s", cm->ctx[g]->device, text.c_str());
}

static bool comm_inject(hw_comm* cm, int g, int phase)
{
    if (g != cm->inject_rank || phase != cm->inject_phase) return false;
    return cm->inject_left.fetch_sub(1, std::memory_order_acq_rel) > 0;
}

// release every rank that may be blocked in a collective: called once, by the first worker that fails in phase 2 (the culprit:
// its rank and text are what the caller is told, whatever errors the abort then causes on the other ranks)
static void comm_abort_all(hw_comm* cm, int culprit_rank, const std::string& text)
{
    if (!cm->fleet.mark_broken(culprit_rank, text)) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (size_t i = 0; i < cm->comm.size(); i++) {
        (void)hipSetDevice(cm->ctx[i]->device);
        if (cm->CommAbort) (void)cm->CommAbort(cm->comm[i]);
    }
    (void)hipSetDevice(cur);
}

// enqueue(g, offset, count, d_out, stream) on every device (by that device's worker), then ONE all-reduce of
// `count_out` doubles per device on the same stream; worker 0 also brings the reduced vector to the host.
// Two phases with a rendezvous between them, see hw_comm: no rank enters the collective unless all ranks enqueued.
template <typename F>
static int comm_run(hw_comm* cm, int64_t n_units, int count_out, double* out, float* wall_ms, F enqueue)
{
    if (count_out > kOutCap) return fail(HW_E_INVALID, "too many outputs");
    if (cm->fleet.broken())
        return fail(HW_E_HIP, "this communicator was aborted after a device failed inside a collective: destroy it and create a new one");
    const int G = (int)cm->ctx.size();
    const auto t0 = std::chrono::steady_clock::now();
    const hw::Fleet::Job job = [&](int g, std::string* text) -> int {
        hw_ctx* c = cm->ctx[g];
        int64_t off, cnt;
        comm_shard(n_units, g, G, &off, &cnt);
        // ---- phase 1: this device's kernels (the only part that can fail on one rank alone)
        int rc = comm_inject(cm, g, 1) ? fail(HW_E_HIP, "injected failure on rank %d before the collective (HWMC_COMM_INJECT_FAIL)", g)
                                       : enqueue(g, off, cnt, cm->host_sum ? c->h_out_dev : c->d_out, c->stream);
        if (rc) *text = g_err;                                  // hw_last_error() is per thread: hand the text to the caller's
        if (!cm->fleet.rendezvous(rc)) {
            // all or nothing: NOBODY enters the collective.  Drain what this device was given and report; the healthy ranks
            // return HW_OK and comm_dispatch hands the failing rank's text to the caller.
            (void)hipStreamSynchronize(c->stream);
            free_retired(c);
            return rc;
        }
        // ---- phase 2: every rank has enqueued its kernels
        hipError_t e = hipSuccess;
        if (comm_inject(cm, g, 2)) {
            rc = fail(HW_E_HIP, "injected failure on rank %d inside the collective phase (HWMC_COMM_INJECT_FAIL)", g);
        } else if (cm->host_sum) {
            e = hipStreamSynchronize(c->stream);               // the column pass stored the sums straight to pinned host memory
            if (e == hipSuccess) memcpy(&cm->host_slots[(size_t)g * kOutCap], c->h_out, count_out * sizeof(double));
        } else {
            // one communicator per thread: no group call needed (each rank's call comes from its own thread, as in the
            // one-process-per-GPU layout)
            const ncclResult_t r = cm->AllReduce(c->d_out, c->d_out, (size_t)count_out, ncclDouble, ncclSum, cm->comm[g], c->stream);
            if (r != ncclSuccess) rc = fail(HW_E_HIP, "RCCL error [ncclAllReduce]: %s", cm->GetErrorString(r));
            if (rc == HW_OK && g == 0) e = hipMemcpyAsync(out, c->d_out, count_out * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (rc == HW_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        // a rank whose calls fail AFTER a peer has aborted the communicators is a victim, not a culprit: the abort makes a
        // healthy rank's all-reduce kernel / stream synchronisation return errors too, and the culprit's text is the one to report
        if ((rc != HW_OK || e != hipSuccess) && cm->fleet.broken()) return HW_OK;
        if (rc == HW_OK && e != hipSuccess) rc = fail(HW_E_HIP, "HIP error behind the collective: %s", hipGetErrorString(e));
        if (rc != HW_OK) {
            *text = g_err;
            if (!cm->host_sum) comm_abort_all(cm, g, g_err);    // the other ranks may be blocked in the all-reduce: release them
                                                                // (HW_COMM_HOST_SUM has no collective: reported, the fleet goes on)
            return rc;
        }
        if (cm->fleet.broken()) return HW_OK;                   // released by a peer's abort: its error is the one reported
        free_retired(c);
        return HW_OK;
    };
    int rc = comm_dispatch(cm, job);
    if (cm->fleet.broken()) {                                   // whatever the lowest failing rank said: the culprit's rank and text
        std::string why;
        const int who = cm->fleet.culprit(&why);
        rc = who >= 0 ? fail(HW_E_HIP, "device %d: %s (the communicator was aborted)", cm->ctx[who]->device, why.c_str())
                      : fail(HW_E_HIP, "a device failed inside the collective; the communicator was aborted");
    }
    if (rc == HW_OK && cm->host_sum)
        for (int k = 0; k < count_out; k++) {
            double t = 0.0;
            for (int g = 0; g < G; g++) t += cm->host_slots[(size_t)g * kOutCap + k];       // rank order: reproducible
            out[k] = t;
        }
    if (wall_ms) *wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// A request one device would refuse is refused HERE, on the calling thread, for all of them, before any device is
// given work (cheap, and the error text names the request, not a rank); what only shows at run time on one device is
// caught by comm_run's rendezvous.
static int comm_check(hw_comm* cm, const hw_zbc_request* r)
{
    for (hw_ctx* c : cm->ctx) {
        int n_steps_S1 = 0;
        const int rc = check_request(c, r, &n_steps_S1);
        if (rc) return rc;
        if (r->stepper == HW_STEPPER_REFERENCE && c->p.precision != HW_F32) return fail(HW_E_INVALID, "the REFERENCE stepper is float32 only");
        if (r->stepper != HW_STEPPER_REFERENCE && r->stepper != HW_STEPPER_SUPERPOSED) return fail(HW_E_INVALID, "unknown stepper %d", r->stepper);
        if (r->drift_mode != HW_DRIFT_BASE && r->drift_mode != HW_DRIFT_SHIFTED) return fail(HW_E_INVALID, "drift_mode must be HW_DRIFT_BASE or HW_DRIFT_SHIFTED");
    }
    return HW_OK;
}

extern "C" {

int hw_comm_create(int32_t n, hw_ctx* const* ctxs, hw_comm** out) { return hw_comm_create_ex(n, ctxs, HW_COMM_RCCL, out); }

int hw_comm_create_ex(int32_t n, hw_ctx* const* ctxs, int32_t flags, hw_comm** out)
{
    if (!out || !ctxs || n < 1 || n > 64) return fail(HW_E_INVALID, "bad argument");
    if (flags != HW_COMM_RCCL && flags != HW_COMM_HOST_SUM) return fail(HW_E_INVALID, "unknown communicator flags %d", flags);
    *out = nullptr;
    hw_comm* cm = new (std::nothrow) hw_comm();
    if (!cm) return fail(HW_E_NOMEM, "out of host memory");
    if (const char* inj = getenv("HWMC_COMM_INJECT_FAIL")) {            // rank:phase[:count] -- rehearsal tests only
        int rk = -1, ph = 0, cnt = 1;
        if (sscanf(inj, "%d:%d:%d", &rk, &ph, &cnt) >= 2 && (ph == 1 || ph == 2)) {
            cm->inject_rank = rk; cm->inject_phase = ph;
            cm->inject_left.store(cnt > 0 ? cnt : 1);
        }
    }
    if (flags == HW_COMM_HOST_SUM) {
        for (int i = 0; i < n; i++) {
            if (!ctxs[i]) { delete cm; return fail(HW_E_INVALID, "null context %d", i); }
            for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) { delete cm; return fail(HW_E_INVALID, "context %d listed twice", i); }
            cm->ctx.push_back(ctxs[i]);
        }
        cm->host_sum = true;
        cm->host_slots.assign((size_t)n * kOutCap, 0.0);
        cm->fleet.start(n, [cm](int g) { (void)hipSetDevice(cm->ctx[g]->device); });
        *out = cm;
        return HW_OK;
    }
    cm->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!cm->lib) cm->lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!cm->lib) { delete cm; return fail(HW_E_HIP, "cannot load RCCL (librccl.so.1): %s", dlerror()); }
    cm->CommInitAll = (decltype(cm->CommInitAll))dlsym(cm->lib, "ncclCommInitAll");
    cm->AllReduce = (decltype(cm->AllReduce))dlsym(cm->lib, "ncclAllReduce");
    cm->GroupStart = (decltype(cm->GroupStart))dlsym(cm->lib, "ncclGroupStart");
    cm->GroupEnd = (decltype(cm->GroupEnd))dlsym(cm->lib, "ncclGroupEnd");
    cm->CommDestroy = (decltype(cm->CommDestroy))dlsym(cm->lib, "ncclCommDestroy");
    cm->CommAbort = (decltype(cm->CommAbort))dlsym(cm->lib, "ncclCommAbort");
    cm->GetErrorString = (decltype(cm->GetErrorString))dlsym(cm->lib, "ncclGetErrorString");
    if (!cm->CommInitAll || !cm->AllReduce || !cm->GroupStart || !cm->GroupEnd || !cm->CommDestroy || !cm->CommAbort || !cm->GetErrorString) {
        delete cm;
        return fail(HW_E_HIP, "RCCL library lacks a required symbol");
    }
    std::vector<int> devs;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) { delete cm; return fail(HW_E_INVALID, "null context %d", i); }
        for (int d : devs) if (d == ctxs[i]->device) { delete cm; return fail(HW_E_INVALID, "device %d used twice", d); }
        cm->ctx.push_back(ctxs[i]);
        devs.push_back(ctxs[i]->device);
    }
    cm->comm.resize(n);
    ncclResult_t r = cm->CommInitAll(cm->comm.data(), n, devs.data());
    if (r != ncclSuccess) {
        const char* msg = cm->GetErrorString(r);
        delete cm;
        return fail(HW_E_HIP, "ncclCommInitAll: %s", msg);
    }
    cm->fleet.start(n, [cm](int g) { (void)hipSetDevice(cm->ctx[g]->device); });
    *out = cm;
    return HW_OK;
}

void hw_comm_destroy(hw_comm* cm)
{
    if (!cm) return;
    cm->fleet.stop();
    for (size_t i = 0; i < cm->comm.size(); i++) {      // (no communicators in HW_COMM_HOST_SUM mode)
        if (cm->fleet.broken()) break;                   // ncclCommAbort has already released them
        (void)hipSetDevice(cm->ctx[i]->device);
        (void)cm->CommDestroy(cm->comm[i]);
    }
    delete cm;   // the library handle stays loaded for the life of the process
}

int hw_comm_zbc_moments(hw_comm* cm, const hw_zbc_request* r, double out[5], float* wall_ms)
{
    if (!cm || !r || !out) return fail(HW_E_INVALID, "null argument");
    if (const int rcq = comm_check(cm, r)) return rcq;
    return comm_run(cm, r->n_pairs, 5, out, wall_ms, [&](int g, int64_t off, int64_t cnt, double* d_out, hipStream_t st) {
        hw_zbc_request q = *r;
        q.n_pairs = cnt;
        q.path_offset = r->path_offset + (uint64_t)off;
        return zbc_enqueue(cm->ctx[g], &q, 1, nullptr, nullptr, d_out, st);
    });
}

int hw_comm_zbc_moments_runs(hw_comm* cm, const hw_zbc_request* r, int32_t n_runs, uint64_t seed_stride, double* out, float* wall_ms)
{
    if (!cm || !r || !out) return fail(HW_E_INVALID, "null argument");
    if (n_runs < 1 || 5 * n_runs > kOutCap) return fail(HW_E_INVALID, "n_runs must be 1..%d", kOutCap / 5);
    if (const int rcq = comm_check(cm, r)) return rcq;
    return comm_run(cm, r->n_pairs, 5 * n_runs, out, wall_ms, [&](int g, int64_t off, int64_t cnt, double* d_out, hipStream_t st) {
        hw_zbc_request q = *r;
        q.n_pairs = cnt;
        q.path_offset = r->path_offset + (uint64_t)off;
        return zbc_runs_enqueue(cm->ctx[g], &q, n_runs, seed_stride, d_out, st);
    });
}

int hw_comm_zcb_curve(hw_comm* cm, int64_t n_pairs, uint64_t seed, double sigma, double* P_sum, float* wall_ms)
{
    if (!cm || !P_sum) return fail(HW_E_INVALID, "null argument");
    if (n_pairs < 0) return fail(HW_E_INVALID, "n_pairs < 0");
    const int n_mat = cm->ctx[0]->p.n_mat;
    for (hw_ctx* c : cm->ctx)
        if (c->p.n_mat != n_mat) return fail(HW_E_INVALID, "the contexts of a communicator must share n_mat");
    int rc = comm_run(cm, n_pairs, n_mat, P_sum, wall_ms, [&](int g, int64_t off, int64_t cnt, double* d_out, hipStream_t st) {
        return zcb_enqueue(cm->ctx[g], cnt, seed, (uint64_t)off, sigma, d_out, st);
    });
    return rc;                                                        // column 0: every shard wrote 2*its count
}

int hw_comm_vega_pathwise(hw_comm* cm, int64_t n_paths, uint64_t seed, double S1, double S2, double K,
                          int32_t antithetic, double out[1], float* wall_ms)
{
    if (!cm || !out) return fail(HW_E_INVALID, "null argument");
    {
        hw_zbc_request q{};
        q.n_pairs = n_paths; q.S1 = S1; q.S2 = S2; q.K = K;
        if (const int rcq = comm_check(cm, &q)) return rcq;
    }
    return comm_run(cm, n_paths, 1, out, wall_ms, [&](int g, int64_t off, int64_t cnt, double* d_out, hipStream_t st) {
        return vega_enqueue(cm->ctx[g], cnt, seed, (uint64_t)off, S1, S2, K, antithetic, d_out, st);
    });
}

}  // extern "C"
