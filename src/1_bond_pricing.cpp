// bin/q1 -- zero-coupon bond curve P(0,T), forward curve f(0,T) by Monte Carlo on MI355X.
// Same surface as the reference's src/1_bond_pricing.cu (stdout sections, data/P.bin, data/f.bin,
// data/q1_results.json, data/P_curve.csv, data/f_curve.csv, data/summary.txt, data/r_paths.bin);
// the work is done by hw_zcb_curve / hw_average_and_forward / hw_sample_paths of the C ABI.
#include "hw_app.hpp"

int main(int argc, char** argv)
{
    const app::Options opt = app::parse(argc, argv);
    const hw_params p = app::make_params(opt);
    const int64_t n = opt.n();
    const int n_mat = p.n_mat, stride = p.n_steps / (n_mat - 1);
    const float spacing = (float)(p.T_final / (n_mat - 1));
    app::ensure_dirs();

    printf("ZERO COUPON BOND PRICING\n");
    printf("Parameters:\n");
    printf("  N_PATHS = %lld (x2 antithetic = %lld effective)\n", (long long)n, (long long)(2 * n));
    printf("  N_STEPS = %d, N_MAT = %d, T = %.1f years\n", p.n_steps, n_mat, p.T_final);
    printf("  a = %.2f, sigma = %.2f, r0 = %.4f\n", p.a, p.sigma, p.r0);
    printf("  theta profile = %s (theta(t>=5) = %.3f + %.3f t)\n\n", opt.profile.c_str(), p.th_a2, p.th_b2);

    app::Fleet fleet;
    fleet.create(p, opt.gpus, opt.device, opt.reduce);
    const uint64_t seed = opt.have_seed ? opt.seed : (uint64_t)time(nullptr);      // 1_bond:53
    printf("RNG: Philox4x32-10, seed %llu, subsequence = path id (no state to initialise)\n", (unsigned long long)seed);

    printf("Running Monte Carlo simulation...\n");
    std::vector<double> P_sum;
    float sim_ms = 0.0f;
    fleet.warm(opt.spinup_ms, [&] { fleet.zcb(n, seed, 0.0, &P_sum, &sim_ms); });   // warm-up launches
    fleet.zcb(n, seed, 0.0, &P_sum, &sim_ms);
    printf("Simulation complete\n");

    std::vector<float> P(n_mat), f(n_mat);
    HW_CHECK(hw_average_and_forward(&p, P_sum.data(), 2 * n, P.data(), f.data()));

    printf("\nRESULTS\n");
    printf("T (years)    P(0,T)         f(0,T)\n");
    for (int i = 0; i < n_mat; i += stride) printf("%5.1f        %.6f       %7.4f%%\n", i * spacing, P[i], f[i] * 100.0f);

    const int last = n_mat - 1;
    printf("\nChecks\n");                                                          // 1_bond:98-104
    printf("P(0,0) = 1.0:      %.6f %s\n", P[0], (P[0] > 0.99f && P[0] < 1.01f) ? "OK" : "ERROR");
    printf("P(0,10) ~ 0.87:    %.6f %s\n", P[last], (P[last] > 0.3f && P[last] < 0.9f) ? "OK" : "ERROR");
    printf("f(0,0) ~ 1.2%%:     %.4f%% %s\n", f[0] * 100.0f, (f[0] > 0.01f && f[0] < 0.02f) ? "OK" : "ERROR");

    const double mpaths = (2.0 * n / sim_ms) / 1000.0;                              // 1_bond:110
    printf("\nPerformance\n");
    printf("Simulation time: %.2f ms\n", sim_ms);
    printf("Effective paths: %lld\n", (long long)(2 * n));
    printf("Throughput: %.2f M paths/sec\n", mpaths);

    app::summary_init("data/summary.txt", p, n);
    printf("\nSaving Results\n");
    app::save_array("data/P.bin", P.data(), n_mat);
    app::save_array("data/f.bin", f.data(), n_mat);
    printf("Results saved.\n");

    app::Json js;
    if (js.open("data/q1_results.json", "Q1: Zero-Coupon Bond Pricing", p, n)) {    // 1_bond:121-138
        js.array("P", P.data(), n_mat);
        js.raw(",\n");
        js.array("f", f.data(), n_mat);
        js.raw(",\n");
        js.performance(sim_ms, 2.0 * n);
        js.raw(",\n");
        fprintf(js.f, "  \"validation\": {\n    \"P_0_0\": %.8f,\n    \"P_0_10\": %.8f,\n    \"f_0_0\": %.8f\n  }\n", P[0], P[last], f[0]);
        js.close();
        printf("Saved data/q1_results.json\n");
    }
    app::csv_timeseries("data/P_curve.csv", "P(0 T)", P.data(), n_mat, spacing);
    app::csv_timeseries("data/f_curve.csv", "f(0 T)", f.data(), n_mat, spacing);

    if (FILE* s = app::summary_section("data/summary.txt", "Q1: ZERO-COUPON BOND PRICING")) {
        fprintf(s, "\nKey Results:\n  P(0,0) = %.8f (expected: 1.0)\n  P(0,10) = %.8f\n  f(0,0) = %.4f%% (expected: ~1.2%%)\n", P[0], P[last], f[0] * 100.0f);
        fprintf(s, "\nPerformance:\n  Simulation time: %.2f ms\n  Throughput: %.2f M paths/sec\n", sim_ms, mpaths);
        fclose(s);
    }

    const int n_show = 32;                                                          // 1_bond:157-168
    std::vector<float> paths((size_t)n_show * (p.n_steps + 1));
    HW_CHECK(hw_sample_paths(fleet.ctx[0], n_show, seed + 1, 0, paths.data()));
    app::save_array("data/r_paths.bin", paths.data(), (int)paths.size());
    fleet.destroy();
    return 0;
}
