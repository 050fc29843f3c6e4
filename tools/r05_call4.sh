#!/bin/bash
# round-5 GPU call: the new / changed GPU tests, then config-4 lane chunk sizes (scratch output under gpurun_out/r05g)
set -o pipefail
O=gpurun_out/r05g
mkdir -p $O
python3 -m pytest tests -m gpu -x -q -k "test_bench_" > $O/gpu_tests.log 2>&1
rc=$?
tail -12 $O/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for c in 4 8 16 32; do
  SR_LANES=2 SR_CHUNK_POLYS=$c python3 bench.py --workload goldilocks_d1048576_b8192 --steps 5 --warmup 2 --no-cpu-baseline --parity-sample 2 > $O/c4_chunk$c.json 2> $O/c4_chunk$c.err || exit 1
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05g/c4_chunk*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(d["ms_per_step"], 2), d["clock_power"]["sclk_mhz"] if "clock_power" in d else None)
PY
