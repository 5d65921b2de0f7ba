#!/usr/bin/env python3
"""Prints the few numbers of a bench.py JSON line that matter when comparing variants."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["valu"]["frac_of_nominal_clock_peak"])
