#!/bin/bash
# A/B of two builds of the library on ONE box: bench.py lines alternating between SR_LIB_PATH=$1 and the in-tree build.
# usage: tools/ab_libs.sh <other .so> <workload> [reps] [extra bench.py args]
other=$1; wl=$2; reps=${3:-3}; shift 3
for r in $(seq $reps); do
  for lib in "$other" ""; do
    if [ -n "$lib" ]; then export SR_LIB_PATH=$lib; tag=other; else unset SR_LIB_PATH; tag=tree; fi
    python bench.py --workload $wl --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', '$wl', '%.1f' % j['value'], '%.4f ms' % j['ms_per_step'])"
  done
done
