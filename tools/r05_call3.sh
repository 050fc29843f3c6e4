#!/bin/bash
# round-5 GPU call 3: Stark clock ramp (warm-up sweep), config-4 split A/B, then the GPU suite (scratch output under gpurun_out/r05c)
set -o pipefail
O=gpurun_out/r05c
mkdir -p $O
for w in 5 50 200 1000; do
  python3 bench.py --workload stark_d4096_b4096 --steps 50 --warmup $w --no-cpu-baseline > $O/stark_w$w.json 2> $O/stark_w$w.err || exit 1
done
for w in 5 200; do
  python3 bench.py --steps 20 --warmup $w --no-cpu-baseline > $O/c2_w$w.json 2> $O/c2_w$w.err || exit 1
  python3 bench.py --workload babybear_d65536_b16384 --steps 20 --warmup $w --no-cpu-baseline > $O/c3_w$w.json 2> $O/c3_w$w.err || exit 1
done
for i in 1 2; do
  python3 bench.py --workload goldilocks_d1048576_b8192 --steps 5 --warmup 2 --no-cpu-baseline --parity-sample 4 > $O/c4_fused_$i.json 2> $O/c4_fused_$i.err || exit 1
  SR_GL_SPLIT_ROWS=1 python3 bench.py --workload goldilocks_d1048576_b8192 --steps 5 --warmup 2 --no-cpu-baseline --parity-sample 4 > $O/c4_split_$i.json 2> $O/c4_split_$i.err || exit 1
done
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
rc=$?
tail -5 $O/gpu_tests.log
exit $rc
