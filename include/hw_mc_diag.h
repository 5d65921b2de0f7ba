/*
 * hw_mc_diag.h -- diagnostic, profiling and tuning entry points of libhw_hip.so.
 *
 * NOT part of the drop-in boundary (include/hw_mc.h): nothing a binding of the reference's three drivers needs is
 * declared here.  Users: bench.py (per-launch event pairs, launch geometry, in-kernel clock), tools/ (sweeps and probes),
 * tests/ (device RNG / exp2 probes against rocRAND, libm and the oracle) and bin/q3's block-size sweep, which restates the
 * reference's cudaFuncGetAttributes / occupancy printout (src/3_sensitivity_analysis.cu:183-239).
 * Same conventions as hw_mc.h (0 or a negative HW_E_* code; hw_last_error()); versioned together with it (HW_ABI_VERSION).
 */
#ifndef HW_MC_DIAG_H
#define HW_MC_DIAG_H

#include "hw_mc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- launch knobs ------------------------------------------------------------------------ */
/* Launch-geometry knobs: threads per block (64..1024, multiple of 64; 0 keeps the current, -1 returns to the
 * automatic choice: 512 when the batch gives every CU >= 4 such workgroups, else 256),
 * grid cap in workgroups per CU (0 = none: one thread slot per pair, the dispatcher hands the workgroups out;
 * with a cap the threads loop), drift-table staging for the REFERENCE stepper (0 scalar loads, 1 LDS), pairs in
 * flight per thread (1 or 2; 0 keeps the current choice, -1 returns to automatic: 2 -- their Philox blocks
 * interleaved -- from TWO chip-fills of pairs up (2^20 on 256 CUs; the Vega kernel: four) in the float engine when a
 * path has at most 1024 steps, else 1).  Environment, read by hw_create: HWMC_SINGLE_LAUNCH=1 finishes the grid-wide sums
 * inside the path kernel (sharded tickets) instead of a second launch; HWMC_CURVE_GENERIC=1 always runs the per-step
 * curve kernel.  None of them changes a result beyond the last bits of the double sums. */
int  hw_set_launch(hw_ctx* ctx, int32_t block_size, int32_t max_blocks_per_cu, int32_t table_in_lds,
                   int32_t pairs_in_flight);

/* ---- kernel resources --------------------------------------------------------------------- */
/* Kernel resource report: replaces the cudaFuncGetAttributes/occupancy printout of
 * run_sensitivity_mc (src/3_sensitivity_analysis.cu:183-239). */
typedef struct hw_kernel_info {
    int32_t vgprs, sgprs, lds_bytes, scratch_bytes, max_threads_per_block;
    int32_t block_size, blocks_per_cu;  /* occupancy for the launch geometry used */
} hw_kernel_info;

int  hw_get_kernel_info(hw_ctx* ctx, const char* kernel /* "zbc" | "zbc_ref" | "zcb" | "zcb_blocked" | "vega" */,
                        hw_kernel_info* out);

/* ---- device probes (tests) ------------------------------------------------------------------ */
/* The first `n_blocks` Philox blocks of subsequence `subseq` as normals, generated ON THE GPU
 * (out[n_blocks*4]); lets tests pin the device RNG against rocRAND and the oracle. */
int  hw_debug_normals(hw_ctx* ctx, uint64_t seed, uint64_t subseq, int32_t n_blocks, float* out);
/* Same blocks as raw 32-bit Philox words (out[n_blocks*4]), bit-exact against rocrand4(). */
int  hw_debug_raw(hw_ctx* ctx, uint64_t seed, uint64_t subseq, int32_t n_blocks, uint32_t* out);

/* out[i] = the double engine's exp2(x[i]) evaluated ON THE GPU (v_exp_f32 of the float-rounded argument + first-order
 * correction, see HW_F64): lets a test pin its error against libm. */
int  hw_debug_exp2(hw_ctx* ctx, const double* x, int32_t n, double* out);

/* ---- profiling hooks (benchmarks) ------------------------------------------------------------- */
/* Per-launch kernel timing for benchmarks: after hw_profile_enable(ctx, n) each *_async / sync
 * hot-path call records a hipEvent pair on its launch stream tightly around the PATH kernel
 * (not the column reduction), for up to n launches.  hw_profile_read synchronises those events
 * and writes the elapsed milliseconds (returns the count via *n_out) and disables profiling. */
int  hw_profile_enable(hw_ctx* ctx, int32_t max_launches);
int  hw_profile_read(hw_ctx* ctx, float* ms, int32_t cap, int32_t* n_out);
/* Start and end of every profiled launch of `ctx`, in milliseconds after the FIRST profiled launch of `ref` started
 * (ref may be ctx; both on the same device): lets a benchmark lay the launches of several streams on one time axis
 * (union of the dispatch intervals).  Call before hw_profile_read, which resets the recording. */
int  hw_profile_spans(hw_ctx* ctx, hw_ctx* ref, float* start_ms, float* end_ms, int32_t cap, int32_t* n_out);
/* The launch geometry the engine would use for `n_units` pairs/paths of kernel "zbc" | "zbc_ref" | "vega" | "zcb" |
 * "zcb_blocked": threads per workgroup, workgroups, iterations per thread, and the bytes of partial sums the path
 * kernel writes (the algorithmic HBM traffic of a launch).  Any output pointer may be NULL. */
int  hw_get_launch_geometry(hw_ctx* ctx, const char* kernel, int64_t n_units, int32_t* block, int32_t* grid,
                            int32_t* iters, int64_t* partial_bytes);
/* Shader clock (MHz) the chip held during the last profiled superposed-ZBC or blocked-curve launch, from in-kernel
 * s_memtime / s_memrealtime probes in up to 64 workgroups spread evenly over the grid (= along the launch, workgroups being
 * dispatched in index order): the time-weighted mean over the probing workgroups' lifetimes (read after the launch completed). */
int  hw_profile_clock(hw_ctx* ctx, double* shader_mhz);
/* The individual probes of that launch: clock (MHz) of each probing workgroup and its entry / exit time in microseconds
 * after the first probe's entry (any output pointer may be NULL); *n_out = how many probed. */
int  hw_profile_clock_samples(hw_ctx* ctx, double* mhz, double* start_us, double* end_us, int32_t cap, int32_t* n_out);

/* Queue q (0 .. n_queues-1) of a pipe: the context whose workspaces its launches use and the index (0..7) of the candidate
 * stream the tuning step picked for it; request i of a batch goes to queue i % n_queues.  With hw_profile_enable /
 * hw_profile_spans on the contexts a test can see that a batch really alternated over distinct streams and that launches of
 * different queues were in flight together, without a wall clock.  The context stays owned by the pipe (never hw_destroy it).
 * Any output pointer may be NULL; q out of range -> HW_E_INVALID. */
int  hw_pipe_queue(hw_pipe* pipe, int32_t q, hw_ctx** ctx_out, int32_t* stream_index_out, int32_t* n_queues_out);

#ifdef __cplusplus
}
#endif
#endif /* HW_MC_DIAG_H */
