#!/usr/bin/env python3
"""Summarise a rocprofv3 run (kernel_stats + FETCH_SIZE/WRITE_SIZE passes) into profiles/<round>/README.md.
usage: make_profile_readme.py profiles/r01"""
import collections
import csv
import os
import sys

d = sys.argv[1]
stats = "kernel_stats_goldilocks_d65536_b16384.csv"
out = []
out.append("# %s -- MI355X (gfx950), workload goldilocks_d65536_b16384 (BASELINE configs[1])\n" % d)
out.append("Commands (on the GPU box, from the repo root; counters in their own passes):\n")
out.append("    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o r1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline")
out.append("    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline")
out.append("    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline\n")
out.append("## kernel_stats (%s)\n" % stats)
# kernel_stats averages over every call, including the tiny launches of bench.py's property gate: the last column is the
# average over the full-batch launches alone, from the kernel trace of the same run
trace = collections.defaultdict(list)
tpath = os.path.join(d, stats.replace("kernel_stats", "kernel_trace"))
if os.path.exists(tpath):
    for r in csv.DictReader(open(tpath)):
        trace[r["Kernel_Name"]].append((int(r["Grid_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out.append("| kernel | calls | avg ms (all calls) | share of GPU time | full-batch launches | avg ms (full-batch) |\n|---|---|---|---|---|---|")
for r in csv.DictReader(open(os.path.join(d, stats))):
    if "sr::" in r["Name"] and "build" not in r["Name"]:
        tr = trace.get(r["Name"], [])
        g = max([x[0] for x in tr], default=0)
        fl = [x[1] for x in tr if x[0] == g]
        out.append("| `%s` | %s | %.3f | %s %% | %d | %.3f |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"],
                                                             float(r["AverageNs"]) / 1e6, r["Percentage"], len(fl),
                                                             sum(fl) / max(1, len(fl)) / 1e6))
agg = {}
for name, cn in (("pmc_FETCH_SIZE", "FETCH_SIZE"), ("pmc_WRITE_SIZE", "WRITE_SIZE")):
    dd = collections.defaultdict(list)
    rows = list(csv.DictReader(open(os.path.join(d, name + ".csv"))))
    full = collections.defaultdict(int)  # only the full-batch launches count (bench.py's property gate also runs tiny ones)
    for r in rows:
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        full[kn] = max(full[kn], int(r["Grid_Size"]))
    for r in rows:
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Grid_Size"]) == full[kn]:
            dd[kn].append(float(r["Counter_Value"]))
    agg[cn] = {k: sum(v) / len(v) for k, v in dd.items()}
out.append("\n## HBM traffic per launch (PMC counters; the CSVs are in KiB)\n")
out.append("FETCH_SIZE is doubled, as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 streaming reads; the calibration")
out.append("point inside the same run is `count_noncanonical_kernel`, which reads exactly 8 GiB (2^30 u64, 8 B per lane) and reports")
cal = [v for k, v in agg["FETCH_SIZE"].items() if "count_noncanonical" in k]
out.append("%.0f KiB.  WRITE_SIZE is taken as is (`fill_uniform_kernel` writes exactly 8 GiB and reports %.0f KiB).\n" % (
    cal[0] if cal else float("nan"), [v for k, v in agg["WRITE_SIZE"].items() if "fill_uniform" in k][0]))
out.append("| kernel | launches/step | FETCH_SIZE x2 (GiB) | WRITE_SIZE (GiB) |\n|---|---|---|---|")
total = 0.0
for k in sorted(agg["FETCH_SIZE"]):
    if "sr::gl::" not in k or "build" in k:
        continue
    per_step = 2 if "cols256_kernel<0>" in k else 1
    f, w = 2 * agg["FETCH_SIZE"][k] / 2**20, agg["WRITE_SIZE"].get(k, 0) / 2**20
    total += per_step * (f + w)
    out.append("| `%s` | %d | %.2f | %.2f |" % (k, per_step, f, w))
out.append("\nWhole step: %.1f GiB of HBM traffic for 16384 ring-muls = %.2f MB per ring-mul = %.2fx the algorithmic 3*D*8 = 1 572 864 B."
           % (total, total * 2**30 / 16384 / 1e6, total * 2**30 / 16384 / 1572864))
out.append("Algorithmic bytes: rows256 kernel 24 GiB per launch (read a, read b, write c); each cols256 launch 16 GiB (read + write one operand).\n")
import json
tags = {"rows": [k for k in agg["FETCH_SIZE"] if "gl::rows256_kernel<2>" in k],
        "fwd_cols": [k for k in agg["FETCH_SIZE"] if "cols256_kernel<0>" in k],
        "inv_cols": [k for k in agg["FETCH_SIZE"] if "cols256_kernel<1>" in k]}
tj = {"workload": "goldilocks_d65536_b16384", "batch": 16384, "unit": "bytes of HBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE)",
      "bytes_per_launch": {t: int(sum(2 * agg["FETCH_SIZE"][k] + agg["WRITE_SIZE"].get(k, 0) for k in ks) * 1024 / max(1, len(ks)))
                           for t, ks in tags.items() if ks}}
json.dump(tj, open(os.path.join(d, "traffic.json"), "w"), indent=1)
# integer-VALU issue roofline inputs: dynamic VALU instructions per wave from the SQ counter pass
sq_path = os.path.join(d, "pmc_SQ_counters.csv")
if os.path.exists(sq_path):
    dd = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = list(csv.DictReader(open(sq_path)))
    full = collections.defaultdict(int)
    for r in rows:
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        full[kn] = max(full[kn], int(r["Grid_Size"]))
    for r in rows:
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Grid_Size"]) == full[kn]:
            dd[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    vj = {"workload": "goldilocks_d65536_b16384", "batch": 16384,
          "note": "SQ_INSTS_VALU / SQ_WAVES per dispatch (rocprofv3 --pmc, own pass); waves per launch = SQ_WAVES",
          "kernels": {}}
    out.append("## VALU instructions per wave (pmc_SQ_counters.csv)\n")
    out.append("| kernel | SQ_WAVES per launch | SQ_INSTS_VALU per wave | SQ_INSTS_SALU per wave |\n|---|---|---|---|")
    for t, ks in tags.items():
        for k in ks:
            if k in dd and "SQ_WAVES" in dd[k]:
                waves = sum(dd[k]["SQ_WAVES"]) / len(dd[k]["SQ_WAVES"])
                valu = sum(dd[k]["SQ_INSTS_VALU"]) / len(dd[k]["SQ_INSTS_VALU"]) / waves
                salu = sum(dd[k].get("SQ_INSTS_SALU", [0])) / max(1, len(dd[k].get("SQ_INSTS_SALU", [0]))) / waves
                vj["kernels"][t] = {"waves_per_launch": int(waves), "valu_per_wave": round(valu, 1)}
                out.append("| `%s` | %d | %.0f | %.0f |" % (k, waves, valu, salu))
    out.append("\nIssue roofline used by bench.py (`integer_valu`): 256 CUs x 4 SIMDs x one wave-instruction per 4 cycles x 2.4 GHz = 6.14e11 wave-instructions/s")
    out.append("(valu_issue_rates_gfx950*.txt: carries, compares, v_cndmask, multiplies, 64-bit ops issue every ~4 cycles per SIMD; plain 32-bit add/logic every ~2).\n")
    json.dump(vj, open(os.path.join(d, "valu.json"), "w"), indent=1)
extra = os.path.join(d, "NOTES.md")
if os.path.exists(extra):
    out.append(open(extra).read())
open(os.path.join(d, "README.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
