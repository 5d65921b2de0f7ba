#!/bin/bash
# Collects the rocprofv3 evidence for a round on the GPU box:  tools/profile_round.sh r01
# kernel trace + stats of the default bench command, then PMC passes (each in its own run).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
export HWMC_BENCH_NO_LIVE_BOUND=1   # bench.py must not start its issue-bound probe (a child process) under the profiler
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --spinup-ms 200 --no-cpu-baseline --no-extras"
echo "rocprofv3 <--kernel-trace --stats | --pmc ...> -- $BENCH" | sed "s#$ROOT/##" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
# the same with the steps strictly one after another on one stream: per-dispatch durations = the kernel alone on the chip
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_s1" -- $BENCH --streams 1 > "$OUT/trace_s1.log" 2>&1
echo "trace_s1 rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY \
          --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
find "$OUT" -name "*.csv" | head -40
