#!/usr/bin/env python3
"""rocprofv3 passes for one bench.py workload, and their summary.

  on the GPU box (counters in their own passes, as MI355X_MICROARCH.md prescribes; the program after `--` is python3 itself):
      python3 tools/profile_workload.py collect <workload> gpurun_out/prof_<workload> [--steps 3]
  anywhere (reads the CSVs the collect step left, writes profiles/<round>/pmc_<workload>.json + kernel_stats + a markdown section):
      python3 tools/profile_workload.py summarize gpurun_out/prof_<workload> <workload> profiles/r02

pmc_<workload>.json carries the source hash of the build it was collected on (bench.py:source_hash); bench.py replays its
numbers only while the hash matches.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PASSES = {
    "trace": ["--kernel-trace", "--stats"],
    "pmc_FETCH_SIZE": ["--pmc", "FETCH_SIZE"],
    "pmc_WRITE_SIZE": ["--pmc", "WRITE_SIZE"],
    "pmc_SQ": ["--pmc", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
               "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS"],
}


def collect(workload, outdir, steps):
    outdir = os.path.abspath(outdir)  # rocprofv3 runs with cwd = /tmp
    os.makedirs(outdir, exist_ok=True)
    for old in glob.glob(os.path.join(outdir, "*.csv")):  # never summarise a previous build's counters as this one's
        os.remove(old)
    # GPU_MAX_HW_QUEUES: bench.py's own os.environ.setdefault runs inside python3, AFTER the profiler's preloaded library has
    # initialised the GPU (with --pmc it has), so the profiled process must find it in its environment already.
    env = dict(os.environ, TMPDIR="/tmp", GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "8"))
    # The library picks its plan itself (sr_plan.lanes = 0) with a probe whose launches have other sizes than the timed ones.  Ask a
    # plain run which plan that is, then pin it (--lanes) in the profiled runs: they carry the timed configuration's launches only.
    lanes = []
    try:
        line = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "1",
                               "--no-cpu-baseline"], env=env, capture_output=True, text=True, cwd="/tmp").stdout
        plan = json.loads([ln for ln in line.splitlines() if ln.startswith("{")][-1])["config"]["plan"]
        if "auto): two lanes" in plan:
            lanes = ["--lanes", "2"]
        elif "auto): one stream" in plan:
            lanes = ["--lanes", "1"]
        print("library plan: %s -> profiled with %s" % (plan, lanes or "no pin (no chunked plan)"), flush=True)
    except (IndexError, ValueError, KeyError):
        print("could not read the library's plan; profiling the default", flush=True)
    for name, flags in PASSES.items():
        d = os.path.join(outdir, name)
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py"),
               "--workload", workload, "--steps", str(steps if name == "trace" else 2), "--warmup", "1", "--no-cpu-baseline"] + lanes
        print("+", " ".join(cmd), flush=True)
        with open(os.path.join(outdir, name + ".bench.json"), "w") as fo, open(os.path.join(outdir, name + ".err"), "w") as fe:
            rc = subprocess.call(cmd, env=env, stdout=fo, stderr=fe, cwd="/tmp")
        if rc:
            print("pass %s failed (rc %d), see %s" % (name, rc, os.path.join(outdir, name + ".err")))
            return rc
        for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
            base = os.path.basename(f)
            kind = "kernel_stats" if "kernel_stats" in base else ("kernel_trace" if "kernel_trace" in base else
                                                                   ("counters" if "counter_collection" in base else None))
            if kind:
                shutil.copy(f, os.path.join(outdir, "%s.%s.csv" % (name, kind)))
        shutil.rmtree(d, ignore_errors=True)
        want = ["kernel_stats", "kernel_trace"] if name == "trace" else ["counters"]
        for kind in want:
            if not os.path.exists(os.path.join(outdir, "%s.%s.csv" % (name, kind))):
                print("pass %s left no %s csv, see %s" % (name, kind, os.path.join(outdir, name + ".err")))
                return 1
    return 0


def tag_of(kernel):
    """bench.py's profile tag of a kernel name (the library brackets the same launches with the same tags)."""
    k = kernel.replace("void ", "")
    if "build" in k or "fill_uniform" in k or "count_noncanonical" in k or "addsub" in k or "__amd" in k:
        return None
    if re.match(r"sr::gl::cols256_pair_kernel<", k):     # both operands' forward passes in one launch (plain column pass)
        return "fwd_cols"
    m = re.match(r"sr::gl::cols256(?:_keep)?_kernel<(\d)", k) or re.match(r"sr::gl::strided\w*_kernel<(?:\d+, )?(\d)", k)
    if m:
        return "fwd_cols" if m.group(1) == "0" else "inv_cols"
    m = re.match(r"sr::rt::(?:cols256|strided)_kernel<sr::\w+, (?:\d+, )?(\d)", k)
    if m:
        return "fwd_cols" if m.group(1) == "0" else "inv_cols"
    m = re.match(r"sr::st::cols_kernel<\d+, (?:\(sr::\w+\))?(\d)", k)
    if m:
        return "fwd_cols" if m.group(1) == "0" else "inv_cols"
    m = re.match(r"sr::cols_kernel<sr::\w+, (?:\(sr::\w+\))?(\d)", k)
    if m:
        return "fwd_cols" if m.group(1) == "0" else "inv_cols"
    if re.match(r"sr::(gl::rows|rt::rows|st::tile_kernel|rows_kernel)", k):
        return "rows"
    return None


MAX_ROWS = 400  # rows of a raw CSV kept in profiles/ (the two-lane plans launch hundreds of chunk kernels per step)


def copy_csv(src, dst):
    """copy a rocprofv3 CSV, keeping the header and the first MAX_ROWS rows (a truncated copy says so in its last line)"""
    with open(src) as fi, open(dst, "w") as fo:
        for i, line in enumerate(fi):
            if i > MAX_ROWS:
                fo.write("# truncated: the summary (pmc_*.json, summary_*.md) was computed from the complete file on the GPU box\n")
                break
            fo.write(line)


def full_batch(rows, grid_key):
    """keep the full-batch launches only (bench.py's property gate also runs a few tiny ones)."""
    rows = [r for r in rows if r.get(grid_key) is not None]  # (the marker line of a truncated copy has no columns)
    full = collections.defaultdict(int)
    for r in rows:
        full[r["Kernel_Name"]] = max(full[r["Kernel_Name"]], int(r[grid_key]))
    return [r for r in rows if int(r[grid_key]) == full[r["Kernel_Name"]]]


def summarize(indir, workload, outdir):
    import bench

    os.makedirs(outdir, exist_ok=True)
    batch = bench.WORKLOADS[workload][2]
    md = ["## %s\n" % workload]
    # kernel durations (trace pass)
    shutil.copy(os.path.join(indir, "trace.kernel_stats.csv"), os.path.join(outdir, "kernel_stats_%s.csv" % workload))
    copy_csv(os.path.join(indir, "trace.kernel_trace.csv"), os.path.join(outdir, "kernel_trace_%s.csv" % workload))
    trace = list(csv.DictReader(open(os.path.join(indir, "trace.kernel_trace.csv"))))
    gk = "Grid_Size_X" if "Grid_Size_X" in trace[0] else "Grid_Size"
    dur = collections.defaultdict(list)
    for r in full_batch(trace, gk):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    md.append("| kernel | tag | full-batch launches | avg ms |\n|---|---|---|---|")
    for kname, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if tag_of(kname) or "addsub" in kname:
            md.append("| `%s` | %s | %d | %.3f |" % (kname.split("(")[0].replace("void ", ""), tag_of(kname), len(v), sum(v) / len(v) / 1e6))
    # counters
    agg = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = list(csv.DictReader(open(os.path.join(indir, "pmc_%s.counters.csv" % cn))))
        dd = collections.defaultdict(list)
        for r in full_batch(rows, "Grid_Size"):
            if r["Counter_Name"] == cn:
                dd[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        agg[cn] = {k: sum(v) / len(v) for k, v in dd.items()}
        copy_csv(os.path.join(indir, "pmc_%s.counters.csv" % cn), os.path.join(outdir, "pmc_%s_%s.csv" % (cn, workload)))
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = list(csv.DictReader(open(os.path.join(indir, "pmc_SQ.counters.csv"))))
    for r in full_batch(rows, "Grid_Size"):
        sq[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    copy_csv(os.path.join(indir, "pmc_SQ.counters.csv"), os.path.join(outdir, "pmc_SQ_%s.csv" % workload))
    # a tag's figures come from its full-batch kernels only: the property gate of bench.py also runs 2-tile launches of the
    # forward-only / inverse-only rows kernels, which carry the same tag
    grid = collections.defaultdict(int)
    for r in rows:
        grid[r["Kernel_Name"]] = max(grid[r["Kernel_Name"]], int(r["Grid_Size"]))
    tag_max = collections.defaultdict(int)
    for kname, g in grid.items():
        if tag_of(kname):
            tag_max[tag_of(kname)] = max(tag_max[tag_of(kname)], g)
    bytes_per_launch, valu = {}, {}
    md.append("\n| kernel | tag | 2 x FETCH_SIZE (GiB) | WRITE_SIZE (GiB) | waves | VALU / wave | SALU / wave |\n|---|---|---|---|---|---|---|")
    # the rows tag of a step is ONE kernel (the fused product); the forward-only / inverse-only rows kernels of bench.py's property
    # gate carry the same tag and, with chunked launches, can have the same grid: keep the one with the most dispatches
    calls = {kname: len(c.get("SQ_WAVES", [])) for kname, c in sq.items()}
    rows_kernels = [kn for kn in agg["FETCH_SIZE"] if tag_of(kn) == "rows" and grid.get(kn, 0) * 4 >= tag_max["rows"]]
    rows_main = max(rows_kernels, key=lambda kn: calls.get(kn, 0)) if rows_kernels else None
    # likewise for the column tags since the forward passes of both operands are one launch of a kernel of its own: the property gate's
    # stand-alone transforms launch the single-operand kernel a few times with the same tag -- listed, not part of the step's figure
    tag_calls = collections.defaultdict(int)
    for kname in agg["FETCH_SIZE"]:
        if tag_of(kname) and grid.get(kname, 0) * 4 >= tag_max[tag_of(kname)]:
            tag_calls[tag_of(kname)] = max(tag_calls[tag_of(kname)], calls.get(kname, 0))
    for kname in sorted(agg["FETCH_SIZE"]):
        t = tag_of(kname)
        if not t or grid.get(kname, 0) * 4 < tag_max[t]:
            continue
        f, w = 2 * agg["FETCH_SIZE"][kname] * 1024, agg["WRITE_SIZE"].get(kname, 0) * 1024  # the CSVs are in KiB
        bytes_per_launch.setdefault(t, []).append(f + w)
        c = sq.get(kname, {})
        waves = sum(c.get("SQ_WAVES", [0])) / max(1, len(c.get("SQ_WAVES", [0])))
        vpw = sum(c.get("SQ_INSTS_VALU", [0])) / max(1, len(c.get("SQ_INSTS_VALU", [0]))) / max(waves, 1)
        spw = sum(c.get("SQ_INSTS_SALU", [0])) / max(1, len(c.get("SQ_INSTS_SALU", [0]))) / max(waves, 1)
        if (t == "rows" and kname != rows_main) or calls.get(kname, 0) * 8 < tag_calls[t]:
            bytes_per_launch[t].pop()  # listed in the table, not part of the step's figure for this tag
        elif waves:
            valu.setdefault(t, []).append((waves, vpw))
        md.append("| `%s` | %s | %.2f | %.2f | %d | %.0f | %.0f |" % (kname.split("(")[0].replace("void ", ""), t, f / 2**30, w / 2**30, waves, vpw, spw))
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    # launches per step and tag of the run the counters come from (bench.py's own line from inside the SQ pass): per-launch figures are
    # only replayed into a run that cuts its batch into the same launches (the same plan)
    lps = None
    try:
        bl = json.load(open(os.path.join(indir, "pmc_SQ.bench.json")))
        lps = bl["roofline"].get("launches_per_step_by_kernel")
    except (OSError, ValueError, KeyError):
        pass
    pj = {"workload": workload, "batch": batch, "source_sha256": bench.source_hash(), "commit": commit, "launches_per_step": lps,
          "unit": "bytes of HBM traffic per launch, 2 * FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE counts half of a streaming read on gfx950); a tag with several passes per transform sums them",
          "bytes_per_launch": {t: int(sum(v)) if t != "rows" and len(v) > 1 else int(sum(v) / len(v)) for t, v in bytes_per_launch.items()},
          "valu": {t: {"waves_per_launch": int(sum(w for w, _ in v) / len(v)), "valu_per_wave": round(sum(x for _, x in v) / len(v), 1)}
                   for t, v in valu.items()}}
    json.dump(pj, open(os.path.join(outdir, "pmc_%s.json" % workload), "w"), indent=1)
    for name in ("trace", "pmc_SQ"):
        src = os.path.join(indir, name + ".bench.json")
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(outdir, "bench_under_%s_%s.json" % (name, workload)))
    open(os.path.join(outdir, "summary_%s.md" % workload), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


def readme(outdir):
    """profiles/<round>/README.md: one table with the roofline fraction of every config, recomputed from the committed files."""
    import bench

    md = ["# %s -- MI355X (gfx950), rocprofv3 summaries and bench lines of the round\n" % outdir,
          "Collected with `python3 tools/profile_workload.py collect <workload> <dir>` on the GPU box (four rocprofv3 passes: `--kernel-trace --stats`,",
          "`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_WAVES SQ_INSTS_VALU ...`, each around `python3 bench.py --workload ... --no-cpu-baseline`),",
          "summarised with `... summarize` / `... readme`.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: half of a streaming read is counted on gfx950);",
          "the CSVs are in KiB.  `pmc_<workload>.json` carries the source hash it was collected at; `bench.py` replays its numbers only while the hash matches.\n",
          "## roofline fraction per config, recomputed\n",
          "`achieved` = bytes the dominant kernel has to move per launch (DESIGN.md section 5; `dominant_kernel_bytes_per_ring_mul` x batch / launches) / its average",
          "duration from the kernel trace of the SAME profile run; `bench` = the `roofline.frac` of the stand-alone bench line in `bench_<name>.json`.\n",
          "| config | workload | dominant kernel | bytes per launch (GB) | trace avg ms | achieved GB/s | frac of 8 TB/s | bench line frac | HBM bytes per launch from PMC (GB) |",
          "|---|---|---|---|---|---|---|---|---|"]
    names = {"goldilocks_d65536_b16384": ("2 (headline)", "default"), "babybear_d65536_b16384": ("3", "babybear"),
             "goldilocks_d1048576_b8192": ("4 (per-GPU shard)", "c4_shard"), "stark_d4096_b4096": ("5", "stark")}
    for w, (cfg, bname) in names.items():
        bj = json.load(open(os.path.join(outdir, "bench_%s.json" % bname)))
        pj = json.load(open(os.path.join(outdir, "pmc_%s.json" % w)))
        r = bj["roofline"]
        batch = bench.WORKLOADS[w][2]
        # the dominant kernel's average duration over the COMPLETE trace: kernel_stats_<workload>.csv (rocprofv3 --stats of the same run;
        # the raw kernel_trace CSV kept beside it is only a sample); of the kernels with the dominant tag, the one with the most time
        spath = os.path.join(outdir, "kernel_stats_%s.csv" % w)
        avg_ms = None
        if os.path.exists(spath):
            rows = [row for row in csv.DictReader(open(spath)) if tag_of(row["Name"]) == r["kernel"]]
            if rows:
                big = max(rows, key=lambda row: float(row["TotalDurationNs"]))
                avg_ms = float(big["AverageNs"]) / 1e6
        per_launch = r["dominant_kernel_bytes_per_ring_mul"] * batch / r["launches_per_step"]
        ach = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms else float("nan")
        md.append("| %s | `%s` | %s | %.2f | %s | %.0f | %.3f | %.3f | %.2f |" % (
            cfg, w, r["kernel"], per_launch / 1e9, ("%.3f" % avg_ms) if avg_ms else "n/a", ach, ach / 8000.0, r["frac"],
            pj["bytes_per_launch"].get(r["kernel"], 0) / 1e9))
    md.append("\nThe trace run and the stand-alone bench run are different launches on (possibly) different boxes: the two fractions agree within the")
    md.append("box-to-box spread (a few percent).  PMC bytes per launch equal the algorithmic bytes of each kernel (no re-reads).")
    md.append("Configs 2, 3 and 4 run their chunks on two internal streams (DESIGN.md section 3): a launch's duration in the trace is time in flight")
    md.append("beside the other stream's kernels, so these per-launch fractions are about half of what the same kernel reaches alone;")
    md.append("`bench_one_stream.json` (`bench.py --lanes 1`: the same kernels on one stream) and `whole_step_frac` are the figures to compare with earlier rounds.\n")
    for w in names:
        sp = os.path.join(outdir, "summary_%s.md" % w)
        if os.path.exists(sp):
            md.append(open(sp).read())
    md.append("## other files\n")
    other = os.path.join(outdir, "OTHER_FILES.md")   # written by hand per round: what the remaining files of the directory are
    if os.path.exists(other):
        md.append(open(other).read())
    open(os.path.join(outdir, "README.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md[:22]))


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
        sys.exit(collect(sys.argv[2], sys.argv[3], steps))
    if sys.argv[1] == "readme":
        readme(sys.argv[2])
        sys.exit(0)
    summarize(sys.argv[2], sys.argv[3], sys.argv[4])
