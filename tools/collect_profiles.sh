#!/bin/bash
# On the GPU box: the rocprofv3 passes per workload, summarised THERE (the raw counter CSVs of the two-lane plans run to hundreds
# of MB); only the summaries and truncated raw copies come back under gpurun_out/profiles_rNN/.
#   gpurun -- 'bash tools/collect_profiles.sh r03 "goldilocks_d65536_b16384 babybear_d65536_b16384 stark_d4096_b4096" onestream'
#   gpurun -- 'bash tools/collect_profiles.sh r03 "goldilocks_d1048576_b8192"'       then   cp gpurun_out/profiles_r03/* profiles/r03/
set -e
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}   # must be in the profiled process's environment before the profiler initialises HIP
rnd=${1:-r03}
workloads=${2:-"goldilocks_d65536_b16384 babybear_d65536_b16384 stark_d4096_b4096 goldilocks_d1048576_b8192"}
root=$(pwd)
out=$root/gpurun_out/profiles_$rnd
mkdir -p $out
for w in $workloads; do
  steps=3; [ $w = goldilocks_d1048576_b8192 ] && steps=1
  python3 tools/profile_workload.py collect $w /tmp/prof_$w --steps $steps
  python3 tools/profile_workload.py summarize /tmp/prof_$w $w $out | tail -3
  rm -rf /tmp/prof_$w
done
if [ "$3" = onestream ]; then
  # the headline workload once more on ONE stream (bench.py --lanes 1 = sr_plan.lanes = 1): exclusive kernel durations
  mkdir -p /tmp/prof_one && cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one/t -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lanes 1 > $out/bench_one_stream_under_trace.json 2> /tmp/prof_one/err.txt
  find /tmp/prof_one -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_goldilocks_d65536_b16384_one_stream.csv \;
fi
ls -la $out | head -40
