#!/bin/bash
# runs every op of valu_energy for 1.5 s under the rocm-smi sampler; prints rate, median power and sclk of the busy samples
bin=${1:-build_tmp/valu_energy}; out=${2:-gpurun_out/energy}; mkdir -p $out
for op in 12 0 1 2 3 4 5 6 7 8 9 10 11; do
  tools/power_trace.sh $out/pw_$op.txt $bin $op 1.5 > $out/rate_$op.txt || exit 1
  python3 - $out/pw_$op.txt $out/rate_$op.txt <<'PY'
import sys, re, statistics
pw, ck = [], []
for ln in open(sys.argv[1]):
    if "card0," not in ln: continue
    f = ln.split("card0,")[1].split(",")
    try:
        c = int(re.sub(r"\D", "", f[4])); w = float(f[8])
    except (ValueError, IndexError):
        continue
    pw.append(w); ck.append(c)
busy = sorted(zip(pw, ck))[len(pw) // 2:]   # upper half of the samples: the loaded ones
print(open(sys.argv[2]).read().strip(), "| median power %.0f W, sclk %d MHz (%d samples)" % (statistics.median(b[0] for b in busy), statistics.median(b[1] for b in busy), len(pw)))
PY
done
