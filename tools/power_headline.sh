mkdir -p gpurun_out/r04z
SR_STEPS=200 bash tools/power_trace.sh gpurun_out/r04z/pw_two.txt python3 tools/bench_lanes_wall.py > gpurun_out/r04z/two.txt 2>&1
SR_STEPS=200 SR_LANES=1 bash tools/power_trace.sh gpurun_out/r04z/pw_one.txt python3 tools/bench_lanes_wall.py > gpurun_out/r04z/one.txt 2>&1
t2=$(grep -o "[0-9.]* ms per batch" gpurun_out/r04z/two.txt | cut -d' ' -f1); t1=$(grep -o "[0-9.]* ms per batch" gpurun_out/r04z/one.txt | cut -d' ' -f1)
echo "two lanes (library default)   $(python3 tools/power_summary.py gpurun_out/r04z/pw_two.txt $t2)" > gpurun_out/r04z/power.txt
echo "one stream (sr_plan.lanes=1)  $(python3 tools/power_summary.py gpurun_out/r04z/pw_one.txt $t1)" >> gpurun_out/r04z/power.txt
cat gpurun_out/r04z/power.txt
