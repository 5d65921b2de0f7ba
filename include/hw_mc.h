/*
 * hw_mc.h -- C ABI of the MI355X-native Hull-White one-factor Monte Carlo engine (libhw_hip.so).
 *
 * This is the drop-in boundary for the reference's hot path
 * (giulialionetti/Monte-Carlo-simulation-of-Hull-White-model-and-sensitivities-computation).
 * The reference has no FFI/plugin interface: its three executables call CUDA kernels directly,
 * with model constants in mutable __constant__ symbols and per-path cuRAND states in device
 * memory.  Each entry point below names the reference call sequence it replaces (file:line,
 * relative to the reference root).  Plain C types only: pointers, sizes, scalars.
 *
 * Conventions
 *   - every function returns 0 on success or a negative HW_E_* code; hw_last_error() gives the text
 *     for the calling thread.  Nothing here calls exit() (the reference's check_cuda does,
 *     include/common.cuh:114-120); the executables in src/ keep the reference's exit(1) behaviour.
 *   - a context is bound to one GPU and one HIP stream; use one context per host thread.
 *   - output buffers are caller-owned and need not be zeroed (the reference requires zeroed
 *     accumulators, src/2_option_pricing.cu:122-126).
 *   - random numbers: Philox4x32-10 in rocRAND's rocrand_state_philox4x32_10 layout,
 *     key = seed, subsequence = global path id (path_offset + local index), offset 0, normals by
 *     rocrand_normal4-style Box-Muller; time step i of a path uses component (i&3) of block (i>>2).
 *     No RNG state lives in memory: "same seed" == common random numbers (the reference copies
 *     48 MB of XORWOW states back and forth for that, src/3_sensitivity_analysis.cu:407-434).
 *   - path counts are int64 and are processed exactly (no rounding up to a block multiple).
 *
 * This header is the STABLE surface -- what INTEGRATION.md binds: lifecycle, market input, the three path kernels with
 * their batched (runs) / multi-scenario forms, sample paths, the host-side post-processing, the two-queue pipe and the
 * multi-GPU communicator.  Tuning knobs, profiling hooks and device probes used by bench.py, tools/ and the tests live
 * in hw_mc_diag.h (same library, same version number); a binding of the reference's drivers does not need them.
 */
#ifndef HW_MC_H
#define HW_MC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HW_ABI_VERSION 6   /* 5: diagnostics, profiling hooks and launch knobs moved to hw_mc_diag.h (same symbols, same library);
                               6: hw_pipe_queue added to hw_mc_diag.h (additive: nothing in this header changed) */

/* error codes */
#define HW_OK            0
#define HW_E_INVALID    -1   /* bad argument */
#define HW_E_HIP        -2   /* HIP runtime error (text in hw_last_error) */
#define HW_E_NO_MARKET  -3   /* hw_set_market not called ("Did you run Q1 first?") */
#define HW_E_NOMEM      -4

/* precision of the per-path stepper and payoff arithmetic.
 * HW_F64 = double STATE and ACCUMULATION with float transcendentals: the normals are float32 Box-Muller values (as in
 * the reference, curand_normal is float) and every exp2 of a discount factor / bond price goes through v_exp_f32 on
 * the float-rounded argument plus a first-order correction (measured: max 8.4e-8 relative, sd 2.6e-8, mean signed error -2.3e-9; bounds pinned by
 * tests/test_gpu_parity.py::test_device_exp2_of_a_double).  It answers "what does float STEPPING cost in accuracy"
 * (BASELINE configs[4]); it is not a correctly-rounded double reference -- the CPU oracle in double is. */
#define HW_F32 0
#define HW_F64 1

/* how the host builds the drift tables */
#define HW_TABLES_F64      0  /* closed-form integrals evaluated in double                  */
#define HW_TABLES_REF_F32  1  /* the reference's float32 host arithmetic, common.cuh:60-84   */

/* which per-step arithmetic the ZBC kernel runs */
#define HW_STEPPER_SUPERPOSED 0 /* r = m(t) +- s*y: deterministic mean path hoisted to the host,
                                   one fma + one add per step and pair (default)              */
#define HW_STEPPER_REFERENCE  1 /* the reference's evolve_hull_white_step verbatim on both legs
                                   (fma + trapezoid per step, drift table read per step)      */

/* drift table selection for a ZBC launch */
#define HW_DRIFT_BASE     0  /* compute_drift_tables(sigma0), common.cuh:60-84                 */
#define HW_DRIFT_SHIFTED  1  /* compute_shifted_drift_table(sigma, sigma0), 3_sens:374-398     */

typedef struct hw_ctx hw_ctx;

/* Replaces the compile-time configuration block include/common.cuh:16-39 and theta_func
 * (common.cuh:228-230).  theta(t) = th_a1 + th_b1 t for t < th_break, th_a2 + th_b2 t after. */
typedef struct hw_params {
    double  a, sigma, r0, T_final;
    int32_t n_steps, n_mat;
    double  th_break, th_a1, th_b1, th_a2, th_b2;
    int32_t precision;    /* HW_F32 | HW_F64 */
    int32_t table_mode;   /* HW_TABLES_F64 | HW_TABLES_REF_F32 */
} hw_params;

/* One ZBC pricing launch == one call of simulate_ZBC_control_variate (common.cuh:286-409)
 * preceded by init_rng (common.cuh:277-280) and followed by the five 4-byte D2H copies
 * (src/2_option_pricing.cu:128-152). */
typedef struct hw_zbc_request {
    int64_t  n_pairs;       /* antithetic pairs == reference N_PATHS threads */
    uint64_t seed;
    uint64_t path_offset;   /* global id of this call's first pair (multi-GPU sharding) */
    double   S1, S2, K;     /* option: exercise, bond maturity, strike */
    double   sigma;         /* volatility for this launch; <= 0 means params.sigma */
    int32_t  drift_mode;    /* HW_DRIFT_BASE | HW_DRIFT_SHIFTED (FD at sigma +- eps) */
    int32_t  stepper;       /* HW_STEPPER_SUPERPOSED | HW_STEPPER_REFERENCE */
} hw_zbc_request;

typedef struct hw_device_info {
    char    name[128];
    int32_t device, compute_units, clock_mhz, wavefront_size;
    int64_t lds_per_cu_bytes, hbm_bytes;
} hw_device_info;

/* ---- lifecycle ------------------------------------------------------------------------- */
int         hw_abi_version(void);
const char* hw_last_error(void);

/* Reference defaults; profile "head" (theta intercept 0.019 as in the mounted source,
 * common.cuh:76,229) or "readme" (0.014, the value behind every published number). */
int  hw_default_params(hw_params* p, const char* profile);

/* Replaces select_gpu() (common.cuh:122-141) + compute_constants() (common.cuh:92-110).
 * device < 0 selects the device with the most free memory, as select_gpu does. */
int  hw_create(const hw_params* p, int device, hw_ctx** out);
void hw_destroy(hw_ctx* ctx);
int  hw_get_device_info(hw_ctx* ctx, hw_device_info* out);
int  hw_get_params(hw_ctx* ctx, hw_params* out);

/* Replaces load_market_data_to_device (common.cuh:171-176).  P, f: n_mat float32 each
 * (the data/P.bin, data/f.bin layout). */
int  hw_set_market(hw_ctx* ctx, const float* P, const float* f, int32_t n_mat);

/* ---- hot path -------------------------------------------------------------------------- */
/* out[5] = {sum X, sum Y, sum X^2, sum Y^2, sum XY} over 2*n_pairs samples (double).
 * kernel_ms (may be NULL): hipEvent time of the launch(es), as 2_option:131-144 measures it.  With kernel_ms = NULL
 * (all synchronous entry points) no event pair is recorded and the call returns the moment its results have landed in
 * pinned host memory (polled completion stamps, ~300 us at most, then hipStreamSynchronize; HWMC_SPIN_WAIT=0: always
 * the latter). */
int  hw_zbc_moments(hw_ctx* ctx, const hw_zbc_request* req, double out[5], float* kernel_ms);

/* Same, enqueue only: results land in d_out (DEVICE pointer, 5 doubles) in stream order on `stream` (a hipStream_t;
 * NULL = the context's stream).  Rules for every *_async call:
 *   - workspaces: a context reuses its device workspaces in stream order.  Calls on ONE stream pipeline freely; a call
 *     on a different stream than the context's previous call first waits, on the device, for the context's earlier
 *     work -- correct, but serial.  For overlap use one context per stream (or hw_pipe_*).
 *   - stream lifetime: `stream` must be alive for the duration of THIS CALL, nothing more.  The library never touches a
 *     caller's stream handle after the call returns (it waits on an event of its own, recorded at the end of the call:
 *     one hipEventRecord per call on a caller's stream, none on the context's own).
 *   - host blocking: none in the steady state; only to grow the workspace (first call at a larger size) or when four
 *     table uploads of this context are still in flight. */
int  hw_zbc_moments_async(hw_ctx* ctx, const hw_zbc_request* req, double* d_out, void* stream);

/* n_runs independent pricings in ONE launch: run k uses seed + k*seed_stride on the same pairs
 * [path_offset, path_offset + n_pairs).  Replaces the 20 malloc / init_rng / launch / 5x D2H cycles of
 * run_zbc_statistical_validation (src/2_option_pricing.cu:225-302); out[n_runs*5], run-major.  Each run's
 * sums are bit-identical to a separate hw_zbc_moments call with that seed.  SUPERPOSED stepper only. */
int  hw_zbc_moments_runs(hw_ctx* ctx, const hw_zbc_request* req, int32_t n_runs, uint64_t seed_stride,
                         double* out, float* kernel_ms);

/* Up to 3 volatility scenarios priced from ONE pass over the same random numbers (common
 * random numbers by construction).  Replaces the three kernel launches + state restores of
 * run_finite_difference (3_sens:400-446): scenario k uses sigma[k] and drift_mode[k].
 * out[n_scen*5].  SUPERPOSED stepper only. */
int  hw_zbc_scenarios(hw_ctx* ctx, const hw_zbc_request* req, int32_t n_scen,
                      const double* sigma, const int32_t* drift_mode,
                      double* out, float* kernel_ms);

/* Replaces init_rng + simulate_zcb (market_data.cuh:25-79): P_sum[n_mat] (double) of
 * exp(-int r) over 2*n_pairs paths at every (n_steps/(n_mat-1))-th step; P_sum[0] = 2*n_pairs.
 * sigma <= 0 means params.sigma. */
int  hw_zcb_curve(hw_ctx* ctx, int64_t n_pairs, uint64_t seed, uint64_t path_offset,
                  double sigma, double* P_sum, float* kernel_ms);
int  hw_zcb_curve_async(hw_ctx* ctx, int64_t n_pairs, uint64_t seed, uint64_t path_offset,
                        double sigma, double* d_P_sum, void* stream);

/* Replaces init_rng + simulate_sensitivity (3_sens:22-96): out[0] = sum over n_paths of
 * (term1 - term2); Vega = out[0] / n_paths.  antithetic = 0 reproduces the reference (one
 * path per draw); 1 also evaluates the mirrored path (then Vega = out[0] / (2 n_paths)). */
int  hw_vega_pathwise(hw_ctx* ctx, int64_t n_paths, uint64_t seed, uint64_t path_offset,
                      double S1, double S2, double K, int32_t antithetic,
                      double out[1], float* kernel_ms);
int  hw_vega_pathwise_async(hw_ctx* ctx, int64_t n_paths, uint64_t seed, uint64_t path_offset,
                            double S1, double S2, double K, int32_t antithetic,
                            double* d_out, void* stream);

/* n_runs independent Vega estimates in ONE launch: run k uses seed + k*seed_stride on the same paths.  Replaces
 * the 20 init_rng / launch / D2H cycles of the Vega bootstrap (src/3_sensitivity_analysis.cu:527-654, seeds
 * base + run*982451653 at :545); out[n_runs].  Each run's sum is bit-identical to a separate hw_vega_pathwise call. */
int  hw_vega_pathwise_runs(hw_ctx* ctx, int64_t n_paths, uint64_t seed, uint64_t path_offset, double S1, double S2,
                           double K, int32_t antithetic, int32_t n_runs, uint64_t seed_stride, double* out,
                           float* kernel_ms);

/* Replaces simulate_paths_show (market_data.cuh:136-160): out[n][n_steps+1] float32 row-major. */
int  hw_sample_paths(hw_ctx* ctx, int32_t n, uint64_t seed, uint64_t path_offset, float* out);

/* ---- a stream of independent pricing calls on ONE GPU, two queues in flight -------------------------
 * (not in the reference: its 20-run validation launches, synchronises and copies 20 times, 2_option:225-302).
 * A pipe owns n_queues contexts and a set of candidate streams.  hw_pipe_zbc_moments enqueues the n requests alternating
 * over the queues -- each one full pricing call, exactly hw_zbc_moments_async -- and synchronises once: a second launch
 * in flight covers the first one's drain and launch gaps.  The sums equal those of n separate
 * hw_zbc_moments calls bit for bit.  wall_ms: host wall time of the call.
 * HIP maps streams onto a few hardware queues unpredictably and two streams that share one serialise: hw_pipe_tune times
 * every pair of candidates on the caller's workload and keeps the fastest.  An untuned pipe tunes inside its first
 * hw_pipe_zbc_moments (on a copy of reqs[0] capped at 2^20 pairs, ~0.1 s, not part of wall_ms).
 * One host thread at a time; a request the engine refuses fails the whole batch before anything is enqueued, and an
 * error surfacing later leaves nothing in flight (the call synchronises the device). */
typedef struct hw_pipe hw_pipe;
int  hw_pipe_create(const hw_params* p, int device, int32_t n_queues /* 1..4 */, hw_pipe** out);
void hw_pipe_destroy(hw_pipe* pipe);
int  hw_pipe_set_market(hw_pipe* pipe, const float* P, const float* f, int32_t n_mat);
/* best_ms_per_call (may be NULL): time per call of the chosen streams over `launches` alternating calls of `sample`. */
int  hw_pipe_tune(hw_pipe* pipe, const hw_zbc_request* sample, int32_t launches /* 0 = 16 */, float* best_ms_per_call);
int  hw_pipe_zbc_moments(hw_pipe* pipe, const hw_zbc_request* reqs, int32_t n, double* out /* [n][5] */, float* wall_ms);

/* ---- several GPUs in one process ---------------------------------------------------------------
 * (not in the reference, which is single-GPU).  A communicator spans contexts created on DIFFERENT devices with identical
 * parameters and market curves.  Each call shards the paths by global id (context g gets a contiguous block); one
 * persistent host thread per device enqueues that device's shard and ONE ncclAllReduce on the device's stream (RCCL over
 * xGMI; librccl.so.1 is dlopen'ed by hw_comm_create) and waits for it.  wall_ms: first launch to reduced result.
 * Errors.  A request is checked against every context before any device is given work.  What only one device can find out
 * at run time is caught by a two-phase job: the all-reduce is entered only if ALL workers enqueued their kernels --
 * otherwise nobody enters it, the call returns "device <id>: <text>" of the failing rank and the communicator stays
 * usable.  A failure inside the collective aborts every communicator (ncclCommAbort): the call returns the error of the
 * rank that failed FIRST (not of a victim of the abort) and later calls on this hw_comm are refused; destroy it.
 * [That abort path has run with ONE rank only -- no multi-GPU box was available; its host protocol is checked with up to
 * 8 workers against a mock collective under ThreadSanitizer, tests/cpp/fleet_probe.cpp.]
 * HWMC_COMM_INJECT_FAIL=rank:phase[:count] (read by hw_comm_create*): failure injection for the rehearsal tests. */
typedef struct hw_comm hw_comm;
int  hw_comm_create(int32_t n, hw_ctx* const* ctxs, hw_comm** out);
/* The same with a choice of how the shards' sums are combined: HW_COMM_RCCL (what hw_comm_create does) or
 * HW_COMM_HOST_SUM -- no RCCL at all: every device's worker thread brings its sums to the host and the caller adds
 * them in rank order (bit-reproducible).  In that mode the contexts may share a device, so the threaded dispatch can
 * be rehearsed with G > 1 on a one-GPU box; it is also the fallback where librccl is not installed. */
#define HW_COMM_RCCL      0
#define HW_COMM_HOST_SUM  1
int  hw_comm_create_ex(int32_t n, hw_ctx* const* ctxs, int32_t flags, hw_comm** out);
void hw_comm_destroy(hw_comm* comm);
int  hw_comm_zbc_moments(hw_comm* comm, const hw_zbc_request* req, double out[5], float* wall_ms);
/* The 20-run validation (hw_zbc_moments_runs) over the fleet: every device prices its shard of the pairs for all n_runs
 * seeds in ONE launch, then ONE all-reduce of 5*n_runs doubles.  out[n_runs*5], run-major. */
int  hw_comm_zbc_moments_runs(hw_comm* comm, const hw_zbc_request* req, int32_t n_runs, uint64_t seed_stride, double* out,
                              float* wall_ms);
int  hw_comm_zcb_curve(hw_comm* comm, int64_t n_pairs, uint64_t seed, double sigma, double* P_sum, float* wall_ms);
int  hw_comm_vega_pathwise(hw_comm* comm, int64_t n_paths, uint64_t seed, double S1, double S2, double K,
                           int32_t antithetic, double out[1], float* wall_ms);

/* ---- host-side pieces of the path (no GPU work) ----------------------------------------- */
/* The tables the launch for (sigma, drift_mode) uses: compute_drift_tables / shifted table.
 * drift, sigma_drift: n_steps doubles each (either may be NULL). */
int  hw_drift_tables(hw_ctx* ctx, double sigma, int32_t drift_mode, double* drift, double* sigma_drift);

/* compute_average_and_forward (market_data.cuh:101-127), float32 like the reference. */
int  hw_average_and_forward(const hw_params* p, const double* P_sum, int64_t n_total,
                            float* P, float* f);

/* recover_theta (src/2_option_pricing.cu:14-35). */
int  hw_recover_theta(const hw_params* p, const float* f, float* theta_rec, float* theta_orig, float* Ts);

/* beta*, control-variate price (2_option:154-175, :277-281), evaluated in double.
 * res[9] = {mean_X, mean_Y, var_Y, cov, beta, adjustment, price, var_X, rho} */
int  hw_cv_price(const double sums[5], int64_t n_total, double P0S2, double res[9]);

/* Exact Hull-White curve for the params' theta on the n_mat grid: P(0,T), f(0,T) as float32
 * (synthetic market input that does not depend on a prior Monte Carlo run). */
int  hw_analytic_curve(const hw_params* p, float* P, float* f);

#ifdef __cplusplus
}
#endif
#endif /* HW_MC_H */
