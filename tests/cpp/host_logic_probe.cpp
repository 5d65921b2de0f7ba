// host_logic_probe.cpp -- exercises the C++ host side of the executables WITHOUT a GPU: the sharding,
// statistics and writers of src/hw_app.hpp and the host-only entry points of the C ABI.  Prints JSON that
// tests/test_host_cpp.py compares with the oracle / NumPy.  Built by the test with g++ (no hipcc needed).
#include <cinttypes>
#include "../../src/hw_app.hpp"

int main(int argc, char** argv)
{
    const char* dir = argc > 1 ? argv[1] : ".";
    hw_params p;
    if (hw_default_params(&p, "readme") != HW_OK) return 2;
    std::vector<float> P(p.n_mat), f(p.n_mat), rec(p.n_mat), orig(p.n_mat), Ts(p.n_mat);
    if (hw_analytic_curve(&p, P.data(), f.data()) != HW_OK) return 3;
    if (hw_recover_theta(&p, f.data(), rec.data(), orig.data(), Ts.data()) != HW_OK) return 4;
    float max_err = 0.0f;
    for (int i = 0; i < p.n_mat; i += 10) max_err = fmaxf(max_err, fabsf(rec[i] - orig[i]));

    // sharding of src/hw_app.hpp Fleet (no contexts needed: fake the vector size)
    app::Fleet fleet;
    fleet.ctx.assign(3, nullptr);
    printf("{\"shards\": [");
    for (int g = 0; g < 3; g++) {
        int64_t off, cnt;
        fleet.shard(1001, g, &off, &cnt);
        printf("%s[%" PRId64 ", %" PRId64 "]", g ? ", " : "", off, cnt);
    }
    fleet.ctx.clear();
    std::vector<double> x;
    for (int i = 0; i < 20; i++) x.push_back(0.0354 + 1e-5 * ((i * 7) % 11));
    const app::RunStats s = app::run_stats(x);
    const double sums[5] = {2.0, 4.0, 3.0, 9.0, 5.0};
    const app::CvResult c = app::cv_price(sums, 2, 1.9);
    printf("], \"stats\": {\"mean\": %.17g, \"sd\": %.17g, \"margin\": %.17g, \"cv_percent\": %.17g},", s.mean, s.sd, s.margin, s.cv_percent);
    printf(" \"cv\": {\"beta\": %.17g, \"price\": %.17g}, \"theta_max_err\": %.9g, \"P100\": %.9g, \"f0\": %.9g}\n", c.beta, c.price, max_err,
           P[100], f[0]);

    // writers: the reference's file formats (output.cuh) into `dir`
    std::string d = dir;
    app::Json js;
    if (!js.open((d + "/probe.json").c_str(), "probe", p, 1 << 20)) return 5;
    js.array("P", P.data(), p.n_mat);
    js.raw(",\n");
    js.performance(0.25f, 2.0 * (1 << 20));
    js.raw("\n");
    js.close();
    app::csv_timeseries((d + "/probe.csv").c_str(), "P(0 T)", P.data(), p.n_mat, 0.1f);
    app::save_array((d + "/probe.bin").c_str(), P.data(), p.n_mat);
    std::vector<float> back(p.n_mat);
    app::load_array((d + "/probe.bin").c_str(), back.data(), p.n_mat);
    return back == P ? 0 : 6;
}
