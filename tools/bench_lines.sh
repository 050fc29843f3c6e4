#!/bin/bash
# The round's bench.py lines on ONE box: the driver's line (plain `python bench.py`), the other BASELINE configs, the variants and the
# multi-rank rehearsals on one GPU.  usage: bash tools/bench_lines.sh gpurun_out/<dir>
out=${1:-gpurun_out/bench_lines}
mkdir -p $out
run() { name=$1; shift; python bench.py "$@" 2>$out/$name.err | grep "^{" > $out/bench_$name.json || { echo "FAILED $name"; tail -5 $out/$name.err; }; }
run default
run default_20 --steps 20 --warmup 5 --no-cpu-baseline
run one_stream --steps 20 --warmup 5 --lanes 1 --no-cpu-baseline
run ntt_rhs --steps 20 --warmup 5 --variant mul_ntt_rhs --no-cpu-baseline
run babybear --steps 10 --warmup 3 --workload babybear_d65536_b16384
run babybear_packed --steps 10 --warmup 3 --workload babybear_d65536_b16384_packed --no-cpu-baseline
run stark --steps 20 --warmup 5 --workload stark_d4096_b4096
run c4_shard --steps 5 --warmup 2 --workload goldilocks_d1048576_b8192
run config0 --workload goldilocks_d1024_b1 --steps 20 --warmup 5
run force_dist --steps 10 --warmup 3 --force-dist --no-cpu-baseline
run 2rank_gloo --gpus 2 --backend gloo --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline
run 4rank_gloo --gpus 4 --backend gloo --batch 2048 --steps 10 --warmup 3 --no-cpu-baseline
for f in $out/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"], 3), "frac", round(r["frac"], 4), "step", round(r["whole_step_frac"], 4),
          "copy", round(r.get("copy_measured", 0)), str(r.get("traffic_source"))[:40], d.get("parity", "")[:60])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
