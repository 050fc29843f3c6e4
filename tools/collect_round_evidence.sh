#!/bin/bash
# Second GPU call of a round's final collection (after tools/collect_profiles.sh; run once profiles/rNN holds the PMC / kernel-stats collection of THIS build, so that bench.py replays it):
# bench lines, Stark stall counters and harness, clock ramp, random soaks, the full GPU suite.  Output: gpurun_out/profiles_r05b/
set -o pipefail
O=gpurun_out/profiles_r05b
mkdir -p $O
bash tools/bench_lines.sh $O > $O/bench_lines.txt 2>&1
python3 tools/pmc_stall_passes.py stark_d4096_b4096 /tmp/stalls_stark > /tmp/stalls_stark.log 2>&1
for p in issue mix lds icache; do echo "== pass $p (rocprofv3 --pmc, per kernel: sum over dispatches, per dispatch) =="; head -40 /tmp/stalls_stark/$p.txt; done > $O/pmc_stalls_stark_d4096_b4096.txt 2>&1
{ echo "# tools/ubench/st_bench.hip: st::tile_kernel<9, MODE_MUL, false>, 4096 ring elements of degree 4096 (32768 tiles), ms per launch";
  echo "# xchg=0: the product kernel; xchg=1: all twelve LDS exchanges removed (wrong results, timing only)";
  for i in 1 2; do build_tmp/st_bench0 4096 30; build_tmp/st_bench1 4096 30; done; } > $O/st_bench.txt 2>&1
for w in 5 50 200 1000; do
  python3 bench.py --workload stark_d4096_b4096 --steps 50 --warmup $w --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d.get('clock_power',{})
print('stark_d4096_b4096 --steps 50 --warmup $w: %.3f ms per step, %.0f ring-muls/s, sclk %.0f MHz, socket %.0f W (%d samples)' % (d['ms_per_step'], d['value'], c.get('sclk_mhz',0), c.get('socket_w',0), c.get('samples',0)))"
done > $O/stark_clock_ramp.txt 2>&1
python3 tools/fuzz_random_parity.py 150 5 > $O/fuzz_random.txt 2>&1
SR_LIB_PATH=$PWD/stark_rings_amd/libstarkrings_hip_check.so python3 tools/fuzz_random_parity.py 150 6 >> $O/fuzz_random.txt 2>&1
python3 -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
rc=$?
tail -4 $O/gpu_tests.log; tail -3 $O/fuzz_random.txt; cat $O/stark_clock_ramp.txt; tail -16 $O/bench_lines.txt
exit $rc
