#!/bin/bash
# Everything a round's profiles/ directory is made of, collected on the GPU box in one go:
#   tools/collect_round.sh r04        (then: python tools/summarize_profile.py gpurun_out/prof_r04 r04, and copy the texts)
# Every step runs under its own timeout; a step that fails stops the script (no further GPU work after a failure).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/collect_$TAG
mkdir -p "$OUT"
cd "$ROOT"
exec 3>&1                             # the script's own stdout: step headers go there, not into the step's output file
step() { echo "== $1" >&3; shift; timeout -k 10 "$@" || { echo "step failed: rc=$?" >&3; exit 1; }; }
step "size scan zbc"  200 python tools/size_scan.py --kernel zbc --rounds 7 > "$OUT/size_scan_zbc.txt" 2>&1
step "size scan zcb"  300 python tools/size_scan.py --kernel zcb --rounds 7 > "$OUT/size_scan_zcb.txt" 2>&1
step "size scan vega" 200 python tools/size_scan.py --kernel vega --rounds 7 > "$OUT/size_scan_vega.txt" 2>&1
step "kernels f32 1000" 100 python tools/time_kernels.py > "$OUT/tk_f32_1000.txt" 2>&1
step "kernels f64 1000" 100 python tools/time_kernels.py --precision f64 > "$OUT/tk_f64_1000.txt" 2>&1
step "kernels f32 4000" 100 python tools/time_kernels.py --grid-steps 4000 > "$OUT/tk_f32_4000.txt" 2>&1
step "kernels f64 4000" 100 python tools/time_kernels.py --grid-steps 4000 --precision f64 > "$OUT/tk_f64_4000.txt" 2>&1
step "stress f32" 200 python tools/time_kernels.py --log2 26 --grid-steps 4000 --reps 3 > "$OUT/stress_f32.txt" 2>&1
step "stress f64" 200 python tools/time_kernels.py --log2 26 --grid-steps 4000 --reps 3 --precision f64 > "$OUT/stress_f64.txt" 2>&1
step "curve event cost" 200 python tools/curve_event_cost.py --P 1 > "$OUT/curve_event_cost_p1.txt" 2>&1
step "readme pin" 200 python tools/readme_pin_report.py > "$OUT/readme_pin.txt" 2>&1
step "mix bound" 100 tools/bin/mix_bound > "$OUT/mix_bound.txt" 2>&1
step "mix bound json" 100 tools/bin/mix_bound --json > "$OUT/mix_bound.json" 2>&1
step "seal cost" 100 python tools/seal_cost_probe.py > "$OUT/seal_cost_probe.txt" 2>&1
step "reference sums" 100 python tools/make_reference_sums.py --check > "$OUT/reference_sums_check.txt" 2>&1
step "shard steps" 300 python tools/shard_steps.py > "$OUT/shard_steps.txt" 2>&1
step "bias probe" 200 python tools/bias_probe.py > "$OUT/bias_probe.txt" 2>&1
step "throughput cpp" 100 ./bin/throughput --theta-profile readme > "$OUT/throughput_cpp.txt" 2>&1
( for k in zbc zbc1 zcb1; do timeout -k 5 60 tools/bin/timeline_probe 20 512 $k || exit 1; done ) > "$OUT/timeline_probe.txt" 2>&1 || { echo "timeline failed"; exit 1; }
step "bench sizes" 600 python tools/bench_sizes.py > "$OUT/bench_size_table.txt" 2>&1
echo "== rocprofv3 passes"
timeout -k 10 900 bash tools/profile_round.sh "$TAG" > "$OUT/profile_round.log" 2>&1 || { echo "profile_round failed"; tail -5 "$OUT/profile_round.log"; exit 1; }
# the PMC passes ran on THIS box: condense them here and hand the summary to the default bench, whose roofline.traffic is then
# a measurement of this box (without HWMC_BENCH_PMC_FILE bench.py reports traffic = null)
python tools/summarize_profile.py "gpurun_out/prof_$TAG" "$TAG" > "$OUT/summarize_profile.log" 2>&1 || { echo "summarize failed"; tail -5 "$OUT/summarize_profile.log"; exit 1; }
mkdir -p "$OUT/profiles" && cp profiles/${TAG}_* "$OUT/profiles/" 2>/dev/null
echo "== default bench (the driver's command)"
HWMC_BENCH_PMC_FILE="profiles/${TAG}_pmc_summary.json" HWMC_BENCH_REGIONS_FILE="$OUT/bench_n1_regions.json" \
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err" || { echo "bench failed rc=$?"; tail -3 "$OUT/bench_n1.err"; exit 1; }
tail -c 2048 "$OUT/bench_n1.json" > "$OUT/bench_n1_tail2k.txt"      # what a record that keeps 2 KB of the line holds: must contain "claims"
grep -q '"claims": {' "$OUT/bench_n1_tail2k.txt" || { echo "the last 2 KB of the line do not hold claims"; exit 1; }
echo "all steps done"
