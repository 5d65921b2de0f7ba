#!/bin/bash
# run_guarded.sh SECONDS LOGFILE cmd...  -- run cmd in the background, poll for its exit, SIGKILL it after SECONDS and
# give up (exit 124) if it cannot be killed, so that a process stuck in the driver never eats the whole gpurun limit.
limit=$1; log=$2; shift 2
"$@" > "$log" 2> "$log.err" &
pid=$!
for ((i = 0; i < limit; i++)); do
    if ! kill -0 $pid 2>/dev/null; then wait $pid; exit $?; fi
    sleep 1
done
echo "run_guarded: $* still running after ${limit}s; state: $(cat /proc/$pid/stat 2>/dev/null | cut -d' ' -f3); stack:" >&2
cat /proc/$pid/wchan 2>/dev/null >&2; echo >&2
kill -9 $pid 2>/dev/null
sleep 3
kill -0 $pid 2>/dev/null && echo "run_guarded: pid $pid survives SIGKILL (state $(cut -d' ' -f3 /proc/$pid/stat))" >&2
exit 124
