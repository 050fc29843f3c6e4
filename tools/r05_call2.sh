#!/bin/bash
# round-5 GPU call 2: harness timings, config-4 split A/B, bench lines, then the GPU suite (scratch output under gpurun_out/r05b)
set -o pipefail
O=gpurun_out/r05b
mkdir -p $O
build_tmp/st_bench0 4096 20 > $O/st_bench.txt 2>&1 && build_tmp/st_bench1 4096 20 >> $O/st_bench.txt 2>&1 && \
build_tmp/st_bench0 16384 10 >> $O/st_bench.txt 2>&1 && build_tmp/st_bench1 16384 10 >> $O/st_bench.txt 2>&1 && \
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err && \
python3 bench.py --workload stark_d4096_b4096 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_stark.json 2> $O/bench_stark.err && \
python3 bench.py --workload stark_d4096_b4096 --batch 16384 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_stark_b16384.json 2> $O/bench_stark_b16384.err && \
for i in 1 2; do
  python3 bench.py --workload goldilocks_d1048576_b8192 --steps 5 --warmup 2 --no-cpu-baseline --parity-sample 4 > $O/c4_fused_$i.json 2> $O/c4_fused_$i.err || exit 1
  SR_GL_SPLIT_ROWS=1 python3 bench.py --workload goldilocks_d1048576_b8192 --steps 5 --warmup 2 --no-cpu-baseline --parity-sample 4 > $O/c4_split_$i.json 2> $O/c4_split_$i.err || exit 1
done && \
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
rc=$?
tail -5 $O/gpu_tests.log
exit $rc
