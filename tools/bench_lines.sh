mkdir -p gpurun_out/r2e
python bench.py --steps 10 --warmup 3 2>gpurun_out/r2e/default.err | grep "^{" > gpurun_out/r2e/bench_default.json || exit 1
python bench.py --steps 10 --warmup 3 --variant mul_ntt_rhs --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r2e/bench_ntt_rhs.json || exit 1
python bench.py --steps 10 --warmup 3 --workload babybear_d65536_b16384 2>/dev/null | grep "^{" > gpurun_out/r2e/bench_babybear.json || exit 1
python bench.py --steps 10 --warmup 3 --workload stark_d4096_b4096 2>/dev/null | grep "^{" > gpurun_out/r2e/bench_stark.json || exit 1
python bench.py --steps 5 --warmup 2 --workload goldilocks_d1048576_b8192 2>/dev/null | grep "^{" > gpurun_out/r2e/bench_c4_shard.json || exit 1
python bench.py --gpus 2 --backend gloo --batch 4096 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r2e/bench_2rank_gloo.json || exit 1
for f in gpurun_out/r2e/bench_*.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"], 3), "frac", round(r["frac"], 4), "step", round(r["whole_step_frac"], 4), "copy", round(r["copy_measured"]), r["traffic_source"][:40], d.get("parity", "")[:60])
PY
done
