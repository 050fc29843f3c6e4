#!/bin/bash
# usage: tools/ab_bench.sh [bench.py args]  -- prints value, ms/step, roofline.frac and the per-kernel milliseconds of one bench line
python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep "^{" | python -c '
import sys, json
d = json.loads(sys.stdin.readline())
r = d["roofline"]
print(round(d["value"]), round(d["ms_per_step"], 3), round(r["frac"], 4), {k: round(v, 3) for k, v in r["per_kernel_ms_per_step"].items()}, "copy", round(r["copy_measured"]))
'
