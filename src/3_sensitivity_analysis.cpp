// bin/q3 -- Vega of the zero-coupon-bond call on MI355X: pathwise derivative, common-random-number
// finite difference, and the "recalibrated" finite difference the reference reports as a diagnostic.
// Same surface as the reference's src/3_sensitivity_analysis.cu: reads data/P.bin, data/f.bin, two y/n
// prompts, writes data/q3_results.json, optional data/vega_bootstrap.csv + data/vega_statistics.txt,
// appends data/summary.txt.
#include "hw_app.hpp"
#include "../include/hw_mc_diag.h"   // hw_get_kernel_info / hw_set_launch: the block-size sweep of run_sensitivity_mc (3_sens:183-239)

static const double kS1 = 5.0, kS2 = 10.0, kEps = 0.001;      // 3_sens:175-177, :404-405

static double cv_price_of(app::Fleet& fleet, const app::Options& opt, int64_t n, uint64_t seed, double sigma,
                          int drift_mode, double P0S2)
{
    hw_zbc_request r{};
    r.n_pairs = n; r.seed = seed; r.S1 = kS1; r.S2 = kS2; r.K = (double)expf(-0.1f);
    r.sigma = sigma; r.drift_mode = drift_mode; r.stepper = opt.stepper;
    double sums[5];
    fleet.zbc(r, sums, nullptr);
    return app::cv_price(sums, 2 * n, P0S2).price;              // run_zbc_price, 3_sens:110-166
}

static void kernel_report(app::Fleet& fleet)
{
    hw_kernel_info ki;
    hw_device_info di;
    HW_CHECK(hw_get_kernel_info(fleet.ctx[0], "vega", &ki));
    HW_CHECK(hw_get_device_info(fleet.ctx[0], &di));
    printf("\nHIP Kernel Analysis (vega_pathwise_kernel, gfx950)\n");            // 3_sens:190-239
    printf("  VGPRs per thread:           %d\n", ki.vgprs);
    printf("  LDS per workgroup:          %d bytes\n", ki.lds_bytes);
    printf("  Scratch per thread:         %d bytes %s\n", ki.scratch_bytes, ki.scratch_bytes ? "SPILLING DETECTED" : "No spilling");
    printf("\n Occupancy Analysis\n");
    printf("  Device:                     %s\n", di.name);
    printf("  CUs:                        %d (4 SIMDs each, wave%d)\n", di.compute_units, di.wavefront_size);
    printf("  Workgroup size:             %d threads\n", ki.block_size);
    printf("  Workgroups per CU:          %d (%d of 32 wave slots)\n", ki.blocks_per_cu, ki.blocks_per_cu * ki.block_size / 64);
    printf("  Theoretical occupancy:      %.1f%%\n\n", 100.0 * ki.blocks_per_cu * ki.block_size / 64 / 32.0);
}

// benchmark_block_sizes, 3_sens:277-371 (1 warm-up + 20 timed launches per size).  The reference sweeps
// {128, 256, 512, 1024} threads per block; 64 (one wavefront) is added because it is the native unit here.
static void block_size_sweep(app::Fleet& fleet, int64_t n, uint64_t seed)
{
    hw_device_info di;
    HW_CHECK(hw_get_device_info(fleet.ctx[0], &di));
    printf("\n---BLOCK SIZE OPTIMIZATION SWEEP---\n\n");
    printf("Testing different workgroup sizes to find optimal configuration.\n\n");
    printf("Block Size | Grid Size | Time (ms) | Throughput (M/s) | Blocks/CU\n");
    printf("-----------|-----------|-----------|------------------|----------\n");
    const int sizes[] = {64, 128, 256, 512, 1024};
    const int n_sizes = (int)(sizeof sizes / sizeof sizes[0]);
    float times[n_sizes];
    int best = 0;
    float best_ms = 1e30f;
    const double K = (double)expf(-0.1f);
    for (int i = 0; i < n_sizes; i++) {
        const int bs = sizes[i];
        HW_CHECK(hw_set_launch(fleet.ctx[0], bs, 0, 0, 0));
        double out;
        float ms = 0.0f, total = 0.0f;
        HW_CHECK(hw_vega_pathwise(fleet.ctx[0], n, seed, 0, kS1, kS2, K, 0, &out, &ms));          // warm-up, 3_sens:307-310
        for (int j = 0; j < 20; j++) {                                                            // 3_sens:313-325
            HW_CHECK(hw_vega_pathwise(fleet.ctx[0], n, seed, 0, kS1, kS2, K, 0, &out, &ms));
            total += ms;
        }
        const float avg = total / 20.0f;
        times[i] = avg;
        const long long nb = (long long)((n + bs - 1) / bs);
        printf("%10d | %9lld | %9.3f | %16.2f | %9.1f\n", bs, nb, avg, (n / avg) / 1000.0, (double)nb / di.compute_units);
        if (avg < best_ms) { best_ms = avg; best = bs; }
    }
    printf("\n--- Analysis ---\nBest configuration: %d threads/block (%.3f ms)\n\nRelative Performance:\n", best, best_ms);
    for (int i = 0; i < n_sizes; i++) {                                                           // 3_sens:349-359
        const float rel = 100.0f * best_ms / times[i];
        printf("  %4d threads/block: %5.1f%%%s%s\n", sizes[i], rel, sizes[i] == best ? " OPTIMAL" : "",
               rel < 90.0f ? " too small, high overhead" : "");
    }
    HW_CHECK(hw_set_launch(fleet.ctx[0], -1, 0, 0, -1));                                          // back to the automatic choice
}

int main(int argc, char** argv)
{
    const app::Options opt = app::parse(argc, argv);
    const hw_params p = app::make_params(opt);
    const int64_t n = opt.n();
    const double K = (double)expf(-0.1f);
    app::ensure_dirs();
    app::Fleet fleet;
    fleet.create(p, opt.gpus, opt.gpus == 1 && opt.device == 0 ? -1 : opt.device, opt.reduce);
    printf("---Question 3: Sensitivity Analysis---\n\n");
    std::vector<float> P, f;
    app::load_market(opt, p, &P, &f);
    fleet.set_market(P.data(), f.data());
    const uint64_t t0 = (uint64_t)time(nullptr);
    const uint64_t seed = opt.have_seed ? opt.seed : t0;                         // 3_sens:713
    const double P0S2 = P[p.n_mat - 1];

    // ---- pathwise derivative (run_sensitivity_mc, 3_sens:169-275) --------------------------
    printf("\n---PATHWISE DERIVATIVE METHOD---\n\n\n");
    printf("Method: Simultaneous simulation of r(t) and d_sigma r(t)\n");
    printf("  Option: ZBC(S1=%.1f, S2=%.1f, K=e^-0.1)\n  Paths:  %lld Monte Carlo simulations\n", kS1, kS2, (long long)n);
    kernel_report(fleet);
    float ms = 0.0f;
    fleet.warm(opt.spinup_ms, [&] { fleet.vega(n, seed, kS1, kS2, K, &ms); });   // warm-up launches
    const double sens_mc = fleet.vega(n, seed, kS1, kS2, K, &ms) / (double)n;    // 3_sens:261
    printf("Performance Results\nVega:   %.6f\nComputation:      %.2f ms\nThroughput:       %.2f M paths/sec\n", sens_mc, ms, (n / ms) / 1000.0);

    printf("\n");
    if (app::ask(opt, "Run block size optimization sweep? (y/n): ")) block_size_sweep(fleet, n, seed);

    // ---- finite difference with common random numbers (run_finite_difference, 3_sens:400-446) --
    printf("\nFINITE DIFFERENCE APPROXIMATION\n\n");
    double sens_fd, p_minus, p_plus;
    {
        hw_zbc_request r{};
        r.n_pairs = n; r.seed = seed + 7; r.S1 = kS1; r.S2 = kS2; r.K = K;
        const double sig[2] = {p.sigma - kEps, p.sigma + kEps};
        const int32_t dm[2] = {HW_DRIFT_SHIFTED, HW_DRIFT_SHIFTED};
        double sums[10] = {0};
        if (fleet.ctx.size() == 1) {                 // both volatilities from ONE pass over the draws
            HW_CHECK(hw_zbc_scenarios(fleet.ctx[0], &r, 2, sig, dm, sums, nullptr));
            p_minus = app::cv_price(sums, 2 * n, P0S2).price;
            p_plus = app::cv_price(sums + 5, 2 * n, P0S2).price;
        } else {                                     // same seed twice == common random numbers
            p_minus = cv_price_of(fleet, opt, n, r.seed, sig[0], HW_DRIFT_SHIFTED, P0S2);
            p_plus = cv_price_of(fleet, opt, n, r.seed, sig[1], HW_DRIFT_SHIFTED, P0S2);
        }
        sens_fd = (p_plus - p_minus) / (2.0 * kEps);                             // 3_sens:443
        printf("  ZBC( sigma - eps) = %.8f\n  ZBC(sigma + eps) = %.8f\n  FD Vega  = %.6f\n", p_minus, p_plus, sens_fd);
    }

    // ---- finite difference with re-simulated curves (run_finite_difference_recalibrated, 3_sens:484-525) --
    printf("\n Running finite differences with market data recalibration (diagnostic; expected to be worse).\n\n");
    double sens_fd_new;
    {
        double pr[2];
        for (int s = 0; s < 2; s++) {
            const double sg = p.sigma + (s ? kEps : -kEps);
            printf("Computing at sigma %c epsilon = %.4f...\n", s ? '+' : '-', sg);
            std::vector<double> P_sum;
            fleet.zcb(n, seed + 11, sg, &P_sum, nullptr);                        // recompute_market_data, 3_sens:449-482
            std::vector<float> Ps(p.n_mat), fs(p.n_mat);
            HW_CHECK(hw_average_and_forward(&p, P_sum.data(), 2 * n, Ps.data(), fs.data()));
            fleet.set_market(Ps.data(), fs.data());
            pr[s] = cv_price_of(fleet, opt, n, seed + 11, sg, HW_DRIFT_BASE, Ps[p.n_mat - 1]);
            printf("  Price = %.8f\n", pr[s]);
        }
        fleet.set_market(P.data(), f.data());                                    // restore, 3_sens:516-520
        sens_fd_new = (pr[1] - pr[0]) / (2.0 * kEps);
        printf("\nrecalibrated Vega: %.6f\n", sens_fd_new);
    }

    // ---- optional 20-run validation + z-test (3_sens:527-695, :740-749) ------------------------
    printf("\n");
    if (opt.answer < 0) { int ch; while ((ch = getchar()) != '\n' && ch != EOF) {} }
    if (app::ask(opt, "Run statistical validation (20 runs)? (y/n): ")) {
        const int runs = 20;
        std::vector<double> v(runs);
        const uint64_t base = opt.have_seed ? opt.seed + 1000003ull : t0 * 2ull;  // 3_sens:539
        printf("Running %d independent Monte Carlo simulations...\n", runs);
        if (fleet.ctx.size() == 1) {                 // all 20 runs in ONE launch, each bit-identical to its own launch
            HW_CHECK(hw_vega_pathwise_runs(fleet.ctx[0], n, base, 0, kS1, kS2, K, 0, runs, 982451653ull, v.data(), nullptr));   // 3_sens:545
            for (int k = 0; k < runs; k++) v[k] /= (double)n;
            printf("  Completed %d/%d runs (one launch)...\n", runs, runs);
        } else {
            for (int k = 0; k < runs; k++) {
                v[k] = fleet.vega(n, base + (uint64_t)k * 982451653ull, kS1, kS2, K, nullptr) / (double)n;   // 3_sens:545
                if ((k + 1) % 5 == 0) printf("  Completed %d/%d runs...\n", k + 1, runs);
            }
        }
        const app::RunStats s = app::run_stats(v);
        printf("\nSTATISTICAL VALIDATION RESULTS\n\nSample Statistics (N = %d runs):\n", runs);
        printf("  Mean Vega:              %.6f\n  Standard Deviation:     %.6f\n  Standard Error:         %.6f\n  Coefficient of Var:     %.4f%%\n\n", s.mean, s.sd, s.se, s.cv_percent);
        printf("95%% Confidence Interval:\n  Lower Bound:            %.6f\n  Upper Bound:            %.6f\n  Margin of Error:        ±%.6f\n  Relative Width:         ±%.4f%%\n\n",
               s.lo, s.hi, s.margin, 100.0 * s.margin / s.mean);
        if (FILE* csv = fopen("data/vega_bootstrap.csv", "w")) {                 // 3_sens:622-630
            fprintf(csv, "run,vega\n");
            for (int k = 0; k < runs; k++) fprintf(csv, "%d,%.8f\n", k + 1, v[k]);
            fclose(csv);
            printf("Saved data/vega_bootstrap.csv\n");
        }
        if (FILE* st = fopen("data/vega_statistics.txt", "w")) {                 // 3_sens:632-653
            fprintf(st, "VEGA ESTIMATE STATISTICAL VALIDATION\n=====================================\n\n");
            fprintf(st, "Monte Carlo Parameters:\n  Paths per run:     %lld\n  Independent runs:  %d\n  Total samples:     %lld\n\n", (long long)n, runs, (long long)(n * runs));
            fprintf(st, "Point Estimate:\n  Mean Vega:         %.6f\n\n", s.mean);
            fprintf(st, "Uncertainty Quantification:\n  Standard Error:    %.6f (%.4f%%)\n  95%% CI:             [%.6f, %.6f]\n\n", s.se, 100.0 * s.se / s.mean, s.lo, s.hi);
            fprintf(st, "Validation:\n  Differences < %.6f are statistically insignificant\n  at the 95%% confidence level (within 2 SE).\n", 2 * s.se);
            fclose(st);
            printf("Saved data/vega_statistics.txt\n");
        }
        // validate_method_agreement, 3_sens:656-695; the reference hard-codes SE = 0.000089, here the measured one is used
        const double se1 = s.sd, z = fabs(sens_mc - sens_fd) / se1;
        printf("\nComparing Pathwise vs Finite Difference:\n  Pathwise Vega:      %.6f\n  Finite Diff Vega:   %.6f\n  Absolute Diff:      %.6f\n  Relative Diff:      %.4f%%\n",
               sens_mc, sens_fd, fabs(sens_mc - sens_fd), 100.0 * fabs(sens_mc - sens_fd) / sens_mc);
        printf("Statistical Test (H0: methods agree):\n  Standard Error (single run): %.6f\n  Z-score:            %.4f\n  Critical value:     1.96 (95%% CI)\n  Result: %s\n\n",
               se1, z, z > 1.96 ? "SIGNIFICANT DIFFERENCE (p < 0.05)" : "NO SIGNIFICANT DIFFERENCE (p > 0.05)");
    }

    // ---- comparison (3_sens:751-809) -------------------------------------------------------------
    printf("Pathwise:        %.6f\n", sens_mc);
    printf("FD (no recalibration):        %.6f  (%.2f%% diff)\n", sens_fd, 100.0 * fabs(sens_mc - sens_fd) / fabs(sens_mc));
    printf("FD (recalibrated):    %.6f  (%.2f%% diff)\n", sens_fd_new, 100.0 * fabs(sens_mc - sens_fd_new) / fabs(sens_mc));
    printf(fabs(sens_mc - sens_fd_new) > fabs(sens_mc - sens_fd) ? "\nRecalibration would make it worse/add no significant benefit.\n"
                                                                  : "\nRecalibration would help reduce model inconsistency.\n");
    const double abs_diff = fabs(sens_mc - sens_fd), rel_pct = 100.0 * abs_diff / fabs(sens_fd);
    printf("\nCOMPARATIVE ANALYSIS\n\n--- Vega Estimates ---\n  Pathwise Derivative (MC):   %.6f\n  Finite Difference (FD):     %.6f\n\n", sens_mc, sens_fd);
    printf("--- Difference Analysis ---\n  Absolute Difference:        %.6f\n  Relative Difference:        %.2f%%\n\n", abs_diff, rel_pct);
    printf("--- Validation ---\n  Sign Check:                 %s\n", (sens_mc > 0 && sens_fd > 0) ? "PASS" : "FAIL");      // 3_sens:785-792
    printf("  Magnitude Check:            %s\n", (sens_mc > 0.05 && sens_mc < 0.5 && sens_fd > 0.05 && sens_fd < 0.5) ? "PASS" : "FAIL");
    printf("Agreement Level: %s difference\n", rel_pct < 10 ? "< 10%" : rel_pct < 25 ? "< 25%" : rel_pct < 50 ? "< 50%" : "> 50%");

    app::Json js;
    if (js.open("data/q3_results.json", "Q3: Sensitivity Analysis", p, n)) {      // 3_sens:812-821
        fprintf(js.f, "  \"results\": {\n    \"sensitivity_mc\": %.6f,\n    \"sensitivity_fd\": %.6f,\n    \"abs_diff\": %.2e\n  }\n", sens_mc, sens_fd, abs_diff);
        js.close();
        printf("Saved data/q3_results.json\n");
    }
    if (FILE* s = app::summary_section("data/summary.txt", "Q3: SENSITIVITY ANALYSIS")) {
        fprintf(s, "  Sens (MC): %.6f\n  Sens (FD): %.6f\n", sens_mc, sens_fd);
        fclose(s);
    }
    fleet.destroy();
    return 0;
}
