mkdir -p gpurun_out/energy
for cfg in "64 8000" "128 4000" "256 2000" "512 1000" "16384 60"; do
  set -- $cfg
  tools/power_trace.sh gpurun_out/energy/res_$1.txt build_tmp/glb_base $1 $2 $1 || exit 1
  python3 - gpurun_out/energy/res_$1.txt <<'PY'
import sys, re, statistics
pw, ck = [], []
for ln in open(sys.argv[1]):
    if "card0," not in ln: continue
    f = ln.split("card0,")[1].split(",")
    try: c = int(re.sub(r"\D", "", f[4])); w = float(f[8])
    except (ValueError, IndexError): continue
    pw.append(w); ck.append(c)
busy = sorted(zip(pw, ck))[len(pw) // 2:]
print("   median power %.0f W, sclk %d MHz (%d samples)" % (statistics.median(b[0] for b in busy), statistics.median(b[1] for b in busy), len(pw)))
PY
done
