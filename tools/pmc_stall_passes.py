#!/usr/bin/env python3
"""Stall / instruction-cache / LDS counter passes for one bench.py workload (development aid).

  python3 tools/pmc_stall_passes.py <workload> <outdir> [--lanes N]

Each pass is `rocprofv3 --pmc <counters> -- python3 bench.py ...` (the program itself after `--`; counters in passes of their own,
no trace flags).  Counter names the box's `rocprofv3 --list-avail` does not know are dropped from a pass instead of failing it.
The per-kernel sums go to <outdir>/<pass>.txt (tools/pmc_by_kernel.py)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "issue": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
              "SQ_INSTS_VALU"],
    "mix": ["SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
            "SQ_ACTIVE_INST_VMEM"],
    "lds": ["SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INST_CYCLES_VMEM",
            "SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_SMEM"],
    "icache": ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQ_IFETCH", "SQ_IFETCH_LEVEL",
               "SQC_DCACHE_REQ", "SQC_DCACHE_MISSES"],
}


def main():
    workload, outdir = sys.argv[1], os.path.abspath(sys.argv[2])
    extra = sys.argv[3:]
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "8"))
    avail = subprocess.run(["rocprofv3", "--list-avail"], env=env, capture_output=True, text=True, cwd="/tmp")
    text = avail.stdout + avail.stderr
    open(os.path.join(outdir, "list_avail.txt"), "w").write(text)
    for name, counters in PASSES.items():
        have = [c for c in counters if c in text] if text.strip() else counters
        print("pass %s: %s (dropped: %s)" % (name, " ".join(have), " ".join(c for c in counters if c not in have) or "none"), flush=True)
        d = os.path.join(outdir, name)
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + have + ["--output-format", "csv", "-d", d, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py"),
               "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
        with open(os.path.join(outdir, name + ".bench.json"), "w") as fo, open(os.path.join(outdir, name + ".err"), "w") as fe:
            rc = subprocess.call(cmd, env=env, stdout=fo, stderr=fe, cwd="/tmp")
        if rc:
            print("pass %s failed (rc %d): %s" % (name, rc, open(os.path.join(outdir, name + ".err")).read()[-600:]), flush=True)
            continue
        with open(os.path.join(outdir, name + ".txt"), "w") as fo:
            subprocess.call([sys.executable, os.path.join(ROOT, "tools", "pmc_by_kernel.py"), d], stdout=fo)
        shutil.rmtree(d, ignore_errors=True)
        print(open(os.path.join(outdir, name + ".txt")).read()[:1500], flush=True)


if __name__ == "__main__":
    main()
