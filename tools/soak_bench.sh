#!/bin/bash
# soak_bench.sh N [bench args...] -- run bench.py N times back to back (each under the run_guarded watchdog), print
# value / ms_per_step of every run and the spread; stops at the first run that fails or has to be killed.
n=${1:-10}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out/soak"
for ((i = 1; i <= n; i++)); do
    "$ROOT/tools/run_guarded.sh" 150 "$ROOT/gpurun_out/soak/run_$i.json" python "$ROOT/bench.py" "$@"
    rc=$?
    if [ $rc -ne 0 ]; then echo "run $i: rc=$rc"; tail -5 "$ROOT/gpurun_out/soak/run_$i.json.err"; exit $rc; fi
    python - "$ROOT/gpurun_out/soak/run_$i.json" $i <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("run %s: value %.1f  ms/step %.5f  serial %.1f  chosen %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["serial"]["value"],
      (d["config"]["stream_selection"] or {}).get("chosen")), flush=True)
PY
done
python - "$ROOT/gpurun_out/soak" $n <<'PY'
import json, sys, statistics
v = [json.loads(open("%s/run_%d.json" % (sys.argv[1], i)).read().strip().splitlines()[-1])["value"] for i in range(1, int(sys.argv[2]) + 1)]
print("n=%d  min %.1f  median %.1f  max %.1f  spread %.2f %%" % (len(v), min(v), statistics.median(v), max(v), 100 * (max(v) - min(v)) / statistics.median(v)))
PY
