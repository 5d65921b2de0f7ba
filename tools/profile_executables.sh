#!/bin/bash
# rocprofv3 kernel statistics of the three executables (q1 -> q2 -> q3 in one scratch directory):
#   tools/profile_executables.sh r01      (then copy the *_kernel_stats.csv files into profiles/)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_exe_$TAG
rm -rf "$OUT"; mkdir -p "$OUT/run"
export TMPDIR=/tmp
cd "$OUT/run"
for q in q1 q2 q3; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$q" -- "$ROOT/bin/$q" --paths-log2 20 --seed 1234 --theta-profile readme --yes > "$OUT/$q.log" 2>&1
    echo "$q rc=$?"
    f=$(find "$OUT/$q" -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${q}_kernel_stats.csv"
done
ls "$OUT"/*.csv
