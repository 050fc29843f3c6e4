#!/bin/bash
# A/B of one plan switch on ONE box: bench.py lines alternating between the default plan and `VAR=VALUE` (parsed by
# stark_rings_amd/_lib.py: plan_from_env; the library itself reads no environment variable).
# usage: tools/ab_env.sh VAR=VALUE <workload> [reps] [extra bench.py args]
kv=$1; wl=$2; reps=${3:-3}; shift 3
for r in $(seq $reps); do
  for mode in default "$kv"; do
    if [ "$mode" = default ]; then envs=""; else envs="$kv"; fi
    env $envs python bench.py --workload $wl --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$mode', '$wl', '%.1f' % j['value'], '%.4f ms' % j['ms_per_step'])"
  done
done
