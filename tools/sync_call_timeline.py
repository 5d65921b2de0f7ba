#!/usr/bin/env python3
"""Where a synchronous hw_zbc_moments call spends its time, from a rocprofv3 kernel trace of bin/throughput:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- <repo>/bin/throughput --theta-profile readme
    python tools/sync_call_timeline.py <dir>
The synchronous calls are the launches whose path kernel starts only after the previous call's column pass has ended (the
pipelined calls overlap).  Per call: path kernel, gap to its column pass, column pass, and the turnaround from the end of the
column pass to the start of the NEXT call's path kernel (host sees the stamps, returns, the caller calls again, launch latency)."""
import csv
import glob
import os
import sys

import numpy as np

src = sys.argv[1]
f = max(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        nm = r["Kernel_Name"]
        kind = "path" if "zbc_superposed_kernel" in nm else "col" if "finalize_columns_kernel" in nm else None
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
calls = []
i = 0
while i + 1 < len(rows):
    if rows[i][2] == "path" and rows[i + 1][2] == "col" and rows[i + 1][0] >= rows[i][1]:
        calls.append((rows[i], rows[i + 1]))
        i += 2
    else:
        i += 1
sync = []
for k in range(1, len(calls) - 1):
    prev_col_end, this_start = calls[k - 1][1][1], calls[k][0][0]
    next_start = calls[k + 1][0][0]
    turn = next_start - calls[k][1][1]
    # a synchronous call: the next launch comes only after the HOST has seen this call's results (2 .. 60 us later); launches
    # queued back to back on one stream (the pipe's tuning trials) follow within a fraction of a microsecond and are not counted
    if this_start >= prev_col_end and 2_000 <= turn < 60_000:
        p, c = calls[k]
        sync.append((p[1] - p[0], c[0] - p[1], c[1] - c[0], next_start - c[1], next_start - p[0]))
a = np.array(sync, dtype=float) / 1e3
print("%d synchronous calls found in %s" % (len(a), os.path.basename(f)))
for name, col in (("path kernel", 0), ("gap before the column pass", 1), ("column pass", 2), ("column pass end -> next call's kernel start", 3), ("call period", 4)):
    print("%-46s median %7.2f us   p10 %7.2f   p90 %7.2f" % (name, np.median(a[:, col]), np.percentile(a[:, col], 10), np.percentile(a[:, col], 90)))
