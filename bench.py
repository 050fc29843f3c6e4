#!/usr/bin/env python3
"""Headline benchmark: batched ring multiplications c = a * b in Fp[X]/(X^D+1) on MI355X.

  python bench.py --gpus N --steps K --warmup W
      N = 1 runs in this process.  N > 1 without a torchrun environment starts N ranks itself (a child
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` of this very script, started
      BEFORE this process touches HIP) and exits with the child's status.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (what the driver runs; same thing)

One "step" = one pass of the hot path (forward NTT x2, slot product, inverse NTT) over one batch of
synthetic coefficient vectors already resident in HBM.  Default workload = BASELINE.json configs[1]:
Goldilocks, D = 2^16, batch = 2^14 per GPU (weak scaling: the batch shards by element, no data-path
collective; the only collective is one RCCL broadcast of the twiddle block at start-up).

  python bench.py --force-dist     N = 1 with the multi-GPU code path switched on: a ONE-rank `nccl` (RCCL) process group, the twiddle
      broadcast on the device view, the rank count and the parity all-reduce on device tensors -- the path the 8-GPU run takes,
      executed on the one GPU at hand.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel, HIP
events on the launch stream) and `cpu_baseline` (the oracle -- a C restatement of the reference's CPU
path -- timed on this host's cores over a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); PyTorch's stream, the context's two
# streams (the lanes of the tuned ring product) and a timing stream need one each, or two of them serialise (DESIGN_APPENDIX.md A.2).
# A host setting for the runtime, made before anything initialises HIP; the library itself reads no environment variable.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    # name: (ring, log2 D, batch per GPU, bytes per coefficient)
    "goldilocks_d65536_b16384": ("goldilocks", 16, 1 << 14, 8),   # BASELINE configs[1] (metric config)
    "babybear_d65536_b16384": ("babybear", 16, 1 << 14, 8),       # configs[2], reference 8-byte layout
    "goldilocks_d1048576_b8192": ("goldilocks", 20, 1 << 13, 8),  # configs[3] per-GPU shard (2^16 / 8)
    "babybear_d65536_b16384_packed": ("babybear", 16, 1 << 14, 4),  # configs[2] on the opt-in packed-u32 boundary (w = 4)
    "stark_d4096_b4096": ("stark", 12, 1 << 12, 32),              # configs[4]
    "goldilocks_d1024_b1": ("goldilocks", 10, 1, 8),              # configs[0] plumbing
    "goldilocks_d1024_b1048576": ("goldilocks", 10, 1 << 20, 8),  # configs[0]'s degree at a GPU-sized batch (informational)
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, -(-q // per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("SR_CPU_THREADS")
    return int(env) if env else n


def source_hash():
    """sha256 over the kernel sources and the C header: the identity of what the .so was built from.  PMC-derived numbers
    committed under profiles/ carry the hash they were measured at and are dropped when it differs."""
    import hashlib

    h = hashlib.sha256()
    src = os.path.join(ROOT, "stark_rings_amd", "csrc")
    for name in sorted(os.listdir(src)) + ["../../include/stark_rings_hip.h"]:
        path = os.path.join(src, name)
        if os.path.isfile(path):
            h.update(name.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


class ClockPowerSampler:
    """Shader clock and socket power of ONE GPU while the timed steps run: a thread of THIS process that reads the amdgpu hwmon files
    (freq1_input = sclk in Hz, power1_input / power1_average = socket power in microwatts) every ~2 ms.  sysfs only: no HIP call, no
    second process, nothing that touches the GPU.  The card is found by the PCI address torch reports for the HIP device."""

    def __init__(self, pci_bus_id):
        import glob
        import threading

        self.files = None
        self.samples = []   # (perf_counter, sclk_hz, power_uw)
        self.card = None
        want = (pci_bus_id or "").lower()
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
            dev = os.path.join(card, "device")
            try:
                addr = os.path.basename(os.path.realpath(dev)).lower()
            except OSError:
                continue
            if want and addr != want:
                continue
            for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                f = os.path.join(hw, "freq1_input")
                pw = [q for q in (os.path.join(hw, "power1_input"), os.path.join(hw, "power1_average")) if os.path.exists(q)]
                if os.path.exists(f) and pw:
                    self.files = (f, pw[0])
                    self.card = "%s (%s)" % (os.path.basename(card), addr)
                    break
            if self.files:
                break
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.files else None

    def start(self):
        if self._thread:
            self._thread.start()
        return self

    def _run(self):
        f_clk, f_pw = self.files
        while not self._stop.is_set():
            try:
                with open(f_clk) as a, open(f_pw) as b:
                    self.samples.append((time.perf_counter(), int(a.read()), int(b.read())))
            except (OSError, ValueError):
                pass
            self._stop.wait(0.002)

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=1.0)

    def window(self, t0, t1):
        """mean sclk (MHz) and socket power (W) over the samples taken in [t0, t1]; None when the box exposes no such files"""
        rows = [(c, w) for (t, c, w) in self.samples if t0 <= t <= t1]
        if not rows:
            return None
        third = max(1, len(rows) // 3)
        mean = lambda xs: sum(xs) / len(xs)
        first, last = rows[:third], rows[-third:]
        return {"sclk_mhz": mean([c for c, _ in rows]) / 1e6, "socket_w": mean([w for _, w in rows]) / 1e6,
                "sclk_mhz_min": min(c for c, _ in rows) / 1e6, "sclk_mhz_max": max(c for c, _ in rows) / 1e6,
                # first against last third of the window: a timed region shorter than the chip's clock / power ramp (a few hundred ms
                # after idle) shows up here -- the hwmon values are the driver's running averages and lag a short burst of load
                "sclk_mhz_first_third": mean([c for c, _ in first]) / 1e6, "sclk_mhz_last_third": mean([c for c, _ in last]) / 1e6,
                "socket_w_first_third": mean([w for _, w in first]) / 1e6, "socket_w_last_third": mean([w for _, w in last]) / 1e6,
                "window_ms": (t1 - t0) * 1e3,
                "samples": len(rows), "source": "amdgpu hwmon freq1_input / power1_input of %s, read by a thread of this process "
                                                "every ~2 ms during the timed steps" % self.card}


def replayed_kernel_stats(workload, dom_kernel_names, here_hash):
    """Average duration (ms) of the dominant kernel in profiles/<latest round>/kernel_stats_<workload>.csv (rocprofv3 --kernel-trace
    --stats of `bench.py --workload ...`), replayed only while the build's source hash equals the one the profile was collected at
    (pmc_<workload>.json of the same collection carries it).  -> (avg_ms, kernel name, source string) or (None, None, reason)."""
    import csv

    try:
        rounds = sorted(r for r in os.listdir(os.path.join(ROOT, "profiles")) if r.startswith("r"))
        for rnd in reversed(rounds):
            stats = os.path.join(ROOT, "profiles", rnd, "kernel_stats_%s.csv" % workload)
            pj = os.path.join(ROOT, "profiles", rnd, "pmc_%s.json" % workload)
            if os.path.exists(stats) and os.path.exists(pj):
                break
        else:
            return None, None, "none: no profiles/*/kernel_stats_%s.csv" % workload
        if json.load(open(pj)).get("source_sha256") != here_hash:
            return None, None, "dropped (stale): %s was collected at another source hash" % os.path.relpath(stats, ROOT)
        rows = [r for r in csv.DictReader(open(stats)) if any(k in r["Name"] for k in dom_kernel_names)]
        if not rows:
            return None, None, "no dominant-kernel row in %s" % os.path.relpath(stats, ROOT)
        big = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        return float(big["AverageNs"]) / 1e6, big["Name"].split("(")[0].replace("void ", ""), \
            "replayed from %s (rocprofv3 --kernel-trace --stats of the same command, source hash %s)" % (os.path.relpath(stats, ROOT), here_hash)
    except (OSError, ValueError, KeyError):
        return None, None, "unreadable"


def self_launch(args):
    """--gpus N > 1 outside torchrun: start the ranks as a child process tree.  Nothing in this process has initialised HIP
    (no torch.cuda call, no library load), so no exec-after-GPU-init can happen; the child is a plain subprocess."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="goldilocks_d65536_b16384", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calibrate", action="store_true",
                    help="after the timed region, also time the one-stream plan (sr_plan.lanes = 1) on a second context and report it as "
                         "roofline.single_stream -- context only, never the headline (off by default: profile runs and the driver's run "
                         "carry the timed configuration's launches only)")
    ap.add_argument("--no-calibration", action="store_true", help="(accepted for older scripts; calibration is off unless --calibrate)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run every collective of the N > 1 path even at N = 1 (a one-rank nccl group)")
    ap.add_argument("--lanes", type=int, default=0, choices=(0, 1, 2),
                    help="sr_plan.lanes: 0 = the library's own choice (default), 1 = one stream, 2 = two lanes (A/B runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--variant", default="mul", choices=("mul", "mul_ntt_rhs"),
                    help="mul: c = a * b, both in coefficient form (the metric).  mul_ntt_rhs: b kept in NTT form (constant operand), "
                         "sr_ring_mul_ntt_rhs_batch_dev -- informational, never the headline")
    ap.add_argument("--parity-sample", type=int, default=-1,
                    help="elements per rank checked bit for bit against the oracle (default: 64 for the D = 2^20 shard, 3 otherwise)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    from stark_rings_amd import CyclotomicRing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs a GPU"
    if world > ndev and args.backend == "nccl" and world > 1:
        raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, %d visible (RCCL refuses two ranks on one device; "
                         "rehearse with --backend gloo)" % (world, world, ndev))
    dev_index = local_rank % ndev  # one rank per GPU on a full node; a 1-GPU rehearsal (gloo) folds ranks onto device 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    sampler = None
    if rank == 0:
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except (AttributeError, RuntimeError):
            bus = None
        sampler = ClockPowerSampler(bus).start()
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # --force-dist outside torchrun: a one-rank rendezvous on a free local port
            import socket

            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    ring_name, k, batch, coeff_bytes = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    d = 1 << k
    from stark_rings_amd._lib import plan_from_env

    plan = plan_from_env({"goldilocks": 0, "babybear": 1, "stark": 2}[ring_name])
    if args.lanes:
        plan.lanes = args.lanes
    ring = CyclotomicRing(ring_name, k, device=dev_index, plan=plan)
    words = batch * ring.words_per_elem
    packed = args.workload.endswith("_packed")  # operands and results are uint32 words (low half of the reference's Fp64 limb)
    if packed and args.variant != "mul":
        raise SystemExit("--variant %s has no packed form" % args.variant)

    # ---- shared twiddles: rank 0's tables broadcast once over RCCL/xGMI, adopted by every rank ----
    bcast_bytes = bcast_ms = ranks_seen = None
    if dist_on:
        from stark_rings_amd.sharding import share_twiddles

        t_b = time.perf_counter()
        bcast_bytes = share_twiddles(ring, dev)  # non-zero ranks zero their block first: the tables really come from rank 0
        bcast_ms = (time.perf_counter() - t_b) * 1e3
        seen = torch.ones(1, dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")  # a DEVICE tensor under RCCL
        dist.all_reduce(seen)  # every rank counted once: the process group really spans `world` ranks
        ranks_seen = int(seen.item())

    # ---- synthetic inputs generated on device (counter-based PRNG; rank-disjoint coefficient ranges) ----
    first = rank * batch * d
    if packed:
        a = torch.empty(words, dtype=torch.int32, device=dev)
        b = torch.empty(words, dtype=torch.int32, device=dev)
        wide = torch.empty(words, dtype=torch.int64, device=dev)  # generator output in the reference layout, narrowed by the library
        for t32, seed in ((a, 0x5EED0001), (b, 0x5EED0002)):
            ring.fill_uniform_dev(wide, seed, first)
            ring.pack32_dev(t32, wide)
        torch.cuda.synchronize()
        del wide

        mul_fn, crt_fn, icrt_fn, add_fn = ring.mul_packed32_dev, ring.elementwise_crt_packed32_dev, ring.elementwise_icrt_packed32_dev, ring.add_packed32_dev
        noncanonical = lambda t: int(((t < 0) | (t >= ring.modulus)).sum().item())
    else:
        a = torch.empty(words, dtype=torch.int64, device=dev)
        b = torch.empty(words, dtype=torch.int64, device=dev)
        ring.fill_uniform_dev(a, 0x5EED0001, first)
        ring.fill_uniform_dev(b, 0x5EED0002, first)
        torch.cuda.synchronize()
        mul_fn, crt_fn, icrt_fn, add_fn = ring.mul_dev, ring.elementwise_crt_dev, ring.elementwise_icrt_dev, ring.add_dev
        noncanonical = ring.count_noncanonical_dev

    # ---- property gate (no oracle here: the oracle is only used inside the cpu_baseline leg below): one untimed step,
    #      outputs canonical, icrt(crt(c)) == c on the result, sampled outputs kept for the cpu_baseline leg's bit-exact check
    wpe = ring.words_per_elem
    if args.variant == "mul_ntt_rhs":
        b_ntt = b.clone()
        ring.elementwise_crt_dev(b_ntt)   # once, outside every timed region: the constant operand lives in NTT form
        step = lambda: ring.mul_ntt_rhs_dev(a, a, b_ntt)
    else:
        step = lambda: mul_fn(a, a, b)
    ring.reserve_scratch(batch)           # no _dev call blocks after this
    step()
    torch.cuda.synchronize()
    # Which plan runs is the LIBRARY's decision (sr_plan.lanes = 0: it timed two lanes against one stream inside reserve_scratch, on this
    # process's real stream-to-queue mapping, and kept the winner); bench.py only reports it.
    plan_now, probe = ring.plan_in_use()
    lanes_now = int(plan_now.lanes)
    if args.lanes:
        plan_used = "sr_plan.lanes = %d given on the command line (A/B run)" % args.lanes
    elif probe:
        plan_used = ("library default (sr_plan.lanes = 0, auto): %s -- its probe took %.2f ms on two lanes against %.2f ms on one stream for %d ring products"
                     " (each plan warmed up and run as it runs a batch)" % ("two lanes" if lanes_now == 2 else "one stream", probe["two_lanes_ms"], probe["one_stream_ms"], probe["elems"]))
    elif lanes_now == 0 and k > 12 and ring_name in ("goldilocks", "babybear"):
        plan_used = "library default (sr_plan.lanes = 0, auto): two lanes, unmeasured -- the batch is below the eight chunks the probe wants"
    else:
        plan_used = "library default (no chunked plan applies to this ring / degree / batch)"
    assert noncanonical(a) == 0, "non-canonical outputs"
    n_sample = args.parity_sample if args.parity_sample >= 0 else (64 if k >= 20 else 3)
    n_sample = min(n_sample, batch)
    sample = sorted({(i * (batch - 1)) // max(n_sample - 1, 1) for i in range(n_sample)}) if n_sample else []
    as_u64 = (lambda t: t.cpu().numpy().view(np.uint32).astype(np.uint64)) if packed else (lambda t: t.cpu().numpy().view(np.uint64).copy())
    sample_out = {e: as_u64(a[e * wpe:(e + 1) * wpe]) for e in sample}
    rt = a[:min(batch, 8) * wpe].clone()
    crt_fn(rt)
    icrt_fn(rt)
    if not torch.equal(rt, a[:rt.numel()]):
        raise SystemExit("PROPERTY FAILURE rank %d: icrt(crt(c)) != c" % rank)
    del rt

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t_end = time.perf_counter()
    elapsed = t_end - t0
    clock_power = sampler.window(t0, t_end) if sampler else None
    rank_ms = None
    if dist_on:  # max over ranks is the job's time; min next to it makes a straggler visible
        cdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_ms = {"min": float(tmin.item()) / args.steps * 1e3, "max": float(t.item()) / args.steps * 1e3}
        elapsed = float(t.item())

    # ---- per-kernel durations over the same K steps, HIP events on the launch stream.  Every PROF_STRIDE-th launch is bracketed
    #      (7: coprime to the 3, 4 or 6 launches of a chunk, so every tag is sampled evenly): bracketing EVERY launch of a two-lane
    #      plan opens gaps in which the other lane's kernel runs alone and reads 10-20 % short (81.7 us printed against
    #      96.9 us inside the timed steps); the sparse sample leaves the step as it runs, which `profile_step_ms` below checks ----
    def profiled_steps(r, fn, stride, steps):
        r.profile_read()                    # drop whatever earlier steps left
        r.profile_enable(stride)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tp0) / steps * 1e3
        pr = r.profile_read()
        r.profile_enable(False)
        return pr, ms

    PROF_STRIDE = 7
    prof, profile_step_ms = profiled_steps(ring, step, PROF_STRIDE, args.steps)
    if any(v["seen"] and not v["launches"] for v in prof.values()):   # too few launches for a sparse sample: bracket every one
        PROF_STRIDE = 1
        prof, profile_step_ms = profiled_steps(ring, step, 1, args.steps)
    for v in prof.values():   # from here on: ms = mean bracketed duration x every launch of the tag, launches = every launch
        v["sampled"] = v["launches"]
        if v["launches"]:
            v["ms"] = v["ms"] / v["launches"] * v["seen"]
        v["launches"] = v["seen"]
    if sampler:
        sampler.stop()
    # The tuned Goldilocks product runs its chunks on two internal streams (DESIGN_APPENDIX.md A.2): the event-timed duration of a launch is
    # then time IN FLIGHT next to the other lane's kernels, not exclusive GPU time.  For a kernel-quality figure that can be compared
    # with earlier rounds the same steps are also profiled on a second context whose plan pins ONE stream (sr_plan.lanes = 1) --
    # a calibration outside the timed region, reported separately as roofline.single_stream, never as the headline.
    prof_single = single_ms = None
    in_flight = sum(v["ms"] for v in prof.values() if v["launches"]) / args.steps
    if rank == 0 and args.variant == "mul" and args.calibrate and not packed and in_flight > 1.2 * (elapsed / args.steps * 1e3):
        plan1 = plan_from_env()
        plan1.lanes = 1
        ring1 = CyclotomicRing(ring_name, k, device=dev_index, plan=plan1)
        ring1.reserve_scratch(batch)
        ring1.mul_dev(a, a, b)
        torch.cuda.synchronize()
        ks = min(args.steps, 5)
        t1 = time.perf_counter()
        for _ in range(ks):
            ring1.mul_dev(a, a, b)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / ks * 1e3
        pr1, _ = profiled_steps(ring1, lambda: ring1.mul_dev(a, a, b), 1, ks)
        prof_single = {t: dict(v, steps=ks) for t, v in pr1.items() if v["launches"]}
        ring1.close()
    # the timed steps kept multiplying a by the (unchanged) b: the final state must still be a batch of canonical ring elements
    # whose transform round-trips -- outputs after the warm-up are checked, not only the first step
    if noncanonical(a) != 0:
        raise SystemExit("PROPERTY FAILURE rank %d: non-canonical outputs after the timed steps" % rank)
    rt = a[-min(batch, 8) * wpe:].clone()
    crt_fn(rt)
    icrt_fn(rt)
    if not torch.equal(rt, a[-rt.numel():]):
        raise SystemExit("PROPERTY FAILURE rank %d: icrt(crt(c)) != c after the timed steps" % rank)
    del rt

    # ---- achievable streaming bandwidth on this box (SURVEY.md 8d asks for the fraction against it as well as against the
    #      8 TB/s spec): a plain device-to-device copy of one operand, read + write counted ----
    copy_gbs = copy_kind = None
    if rank == 0:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def stream_rate(fn, nbytes):
            fn()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(4):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            return 4 * nbytes / (ev0.elapsed_time(ev1) * 1e-3) / 1e9

        nb = a.numel() * a.element_size()
        rates = {"hipMemcpy device-to-device (1 read + 1 write stream)": stream_rate(lambda: b.copy_(a), 2 * nb),
                 "element-wise ring add a += b (2 read + 1 write streams)": stream_rate(lambda: add_fn(b, a), 3 * nb)}
        copy_kind = max(rates, key=rates.get)
        copy_gbs = rates[copy_kind]

    # ---- parity of this rank's sample against the oracle (CHECKER only, outside every timed region; every rank checks its own
    #      shard: SURVEY.md 8d asks for >= 64 sampled elements per GPU at D = 2^20) ----
    parity = None
    if not args.no_cpu_baseline and sample:
        import oracle_lib as O  # test infrastructure: used as the checker here and as the timed CPU baseline below, never as product

        F = O.FIELD_ID[ring_name]
        threads = max(1, usable_cores() // world)
        ea = np.concatenate([O.fill_uniform(F, 0x5EED0001, first + e * d, d) for e in sample])
        eb = np.concatenate([O.fill_uniform(F, 0x5EED0002, first + e * d, d) for e in sample])
        want = O.pow2_ring_mul(F, ea, eb, k, len(sample), threads)
        bad = [e for i, e in enumerate(sample) if not np.array_equal(sample_out[e], want[i * wpe:(i + 1) * wpe])]
        ok = torch.tensor([0 if bad else 1], dtype=torch.int64, device=dev if (dist_on and args.backend == "nccl") else "cpu")
        if dist_on:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if bad:
            print("PARITY FAILURE rank %d: elements %s differ from the oracle" % (rank, bad[:8]), file=sys.stderr)
        if int(ok.item()) != 1:
            raise SystemExit("PARITY FAILURE: GPU result differs from the oracle on at least one rank")
        parity = ("bit-exact vs oracle on %d sampled elements per rank (%d ranks) of the first step; all outputs canonical; "
                  "icrt(crt(c)) == c" % (len(sample), world))
        del ea, eb, want

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    total_muls = world * batch * args.steps
    value = total_muls / elapsed
    bytes_per_mul = 3 * d * coeff_bytes  # read a, read b, write c
    kern = {t: v for t, v in prof.items() if v["launches"]}
    # dominant kernel = the longest single launch of the step.  Since the forward column passes of both operands are ONE launch, that
    # launch and the rows launch take about equally long at D = 2^16 (85 against 78 us; they did in total before): the rows kernel --
    # the one that computes the most per ring element, and the lower of the two fractions -- stays the subject unless another launch is
    # clearly (> 25 %) longer, so that the tag does not flip between runs; `by_kernel` carries every tag's own figure.
    per_launch = {t: kern[t]["ms"] / kern[t]["launches"] for t in kern}
    dom_tag = max(per_launch, key=per_launch.get)
    if "rows" in per_launch and per_launch["rows"] >= 0.8 * per_launch[dom_tag]:
        dom_tag = "rows"
    dom = kern[dom_tag]
    launches_per_step = dom["launches"] / args.steps
    dom_avg_ms = dom["ms"] / dom["launches"]
    # algorithmic bytes of ONE LAUNCH of the dominant kernel = what that kernel itself has to move per ring element (DESIGN.md
    # section 5): the fused rows kernel reads a, b and writes c (3 D w'); a column pass reads and writes one operand (2 D w').
    # w' = bytes per coefficient of the buffers THAT kernel touches: 8 (Goldilocks), 32 (Stark), and 4 for BabyBear, whose
    # kernels between the boundary passes work on the library's packed 32-bit scratch (the 8-byte boundary words appear only
    # in the column passes: read 8, write 4 forward; read 4, write 8 inverse).
    wk = {"goldilocks": 8, "babybear": 4, "stark": 32}[ring_name]
    if dom_tag == "rows":
        kernel_bytes_per_elem = 3 * d * wk
    elif ring_name == "babybear":
        kernel_bytes_per_elem = d * (coeff_bytes + wk)  # (packed boundary: 4 + 4)
    else:
        kernel_bytes_per_elem = 2 * d * wk
    if k <= 12 and ring_name != "stark" or launches_per_step == 0:
        kernel_bytes_per_elem = bytes_per_mul  # one fused launch at the boundary layout
    fwd_operands = 2 if args.variant == "mul" else 1   # operands whose forward column pass runs in a step (one launch or two)
    alg_bytes_per_launch = kernel_bytes_per_elem * batch / max(launches_per_step, 1e-9) * (fwd_operands if dom_tag == "fwd_cols" else 1)
    # the same figure for every tag of the step (multi-launch degrees excepted: one fused launch has no separate column tags)
    by_kernel = {}
    if not (k <= 12 and ring_name != "stark"):
        for t in kern:
            per_elem = 3 * d * wk if t == "rows" else (d * (coeff_bytes + wk) if ring_name == "babybear" else 2 * d * wk)
            bpl = per_elem * batch / (kern[t]["launches"] / args.steps) * (fwd_operands if t == "fwd_cols" else 1)
            by_kernel[t] = {"avg_ms": per_launch[t], "algorithmic_bytes_per_launch": bpl,
                            "frac": bpl / (per_launch[t] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    achieved_gbs = alg_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9
    step_gbs = bytes_per_mul * batch * world / (elapsed / args.steps) / 1e9 / world

    # HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own passes, FETCH_SIZE doubled per MI355X_MICROARCH.md) and
    # dynamic VALU instructions per wave (SQ counter pass) are NOT measured in this run: they are replayed from
    # profiles/<round>/pmc_<workload>.json, which records the source hash of the build it was collected on.  A different
    # hash (any kernel edit since) drops them instead of presenting stale numbers as current.
    traffic = step_traffic = valu = valu_step = valu_scaled = None
    traffic_source = "none: no profiles/*/pmc_%s.json" % args.workload
    try:
        rounds = sorted(r for r in os.listdir(os.path.join(ROOT, "profiles")) if r.startswith("r"))
        for rnd in reversed(rounds):
            pj_path = os.path.join(ROOT, "profiles", rnd, "pmc_%s.json" % args.workload)
            if os.path.exists(pj_path):
                break
        pj = json.load(open(pj_path))
        here = source_hash()
        lps_now = {t: kern[t]["launches"] / args.steps for t in kern}
        lps_then = pj.get("launches_per_step")
        if pj.get("source_sha256") == here and lps_then and pj.get("batch") and all(t in pj.get("valu", {}) for t in lps_then):
            # VALU wave-instructions per RING ELEMENT are a property of the kernels, not of the batch: a run at another batch (or another
            # cut into launches) scales them by its own element count; the exact per-launch form below overrides when everything matches
            per_elem = sum(pj["valu"][t]["waves_per_launch"] * pj["valu"][t]["valu_per_wave"] * lps_then[t] for t in lps_then) / pj["batch"]
            valu_step = per_elem * batch
            valu_scaled = "per ring element from the profile's batch of %d" % pj["batch"]
        if pj.get("batch") != batch:
            traffic_source = "dropped: %s was collected at batch %s" % (os.path.relpath(pj_path, ROOT), pj.get("batch"))
        elif lps_then is None or any(abs(lps_then.get(t, -1) - v) > 1e-6 for t, v in lps_now.items()):
            # per-launch bytes and waves only mean something for launches of the same size: another plan (one stream against two
            # lanes, another chunk) cuts the batch into other launches
            traffic_source = "dropped (other plan): %s was collected with %s launches per step, this run has %s" % (
                os.path.relpath(pj_path, ROOT), lps_then, lps_now)
        elif pj.get("source_sha256") != here:
            traffic_source = "dropped (stale): %s was collected at source hash %s, this build is %s" % (
                os.path.relpath(pj_path, ROOT), pj.get("source_sha256"), here)
        else:
            traffic_source = "replayed from %s (source hash %s, commit %s); not measured in this run" % (
                os.path.relpath(pj_path, ROOT), here, pj.get("commit"))
            traffic = pj["bytes_per_launch"].get(dom_tag)
            if all(t in pj["bytes_per_launch"] for t in kern):
                step_traffic = sum(pj["bytes_per_launch"][t] * kern[t]["launches"] / args.steps for t in kern)
            if all(t in pj.get("valu", {}) for t in kern):
                # VALU wave-instructions of ONE STEP: every tag's waves per launch x VALU per wave x launches per step
                valu_step = sum(pj["valu"][t]["waves_per_launch"] * pj["valu"][t]["valu_per_wave"] * kern[t]["launches"] / args.steps for t in kern)
                valu_scaled = None
            if dom_tag in pj.get("valu", {}):
                kv = pj["valu"][dom_tag]
                rate = kv["waves_per_launch"] * kv["valu_per_wave"] / (dom_avg_ms * 1e-3)
                peak = 256 * 4 * 2.4e9 / 4
                valu = {"bound": "valu-issue", "achieved": rate, "peak": peak, "unit": "wave-instructions/s", "frac": rate / peak,
                        "valu_instructions_per_wave": kv["valu_per_wave"], "waves_per_launch": kv["waves_per_launch"],
                        "note": "the DOMINANT kernel's in-flight issue rate beside the other lane's kernels, nominal 2.4 GHz peak -- supplementary; "
                                "whole_step below is the figure that says how close the step is to its issue floor"}
    except (OSError, ValueError, KeyError, IndexError, NameError):
        pass

    # ---- the figures a reader of this ONE line needs to see why the whole step is where it is ----
    # (a) the dominant kernel's duration a second way: rocprofv3's own average for it over the same command, replayed from profiles/
    #     while the source hash matches; roofline.frac is the LOWER of the two fractions
    dom_names = {"rows": ("rows256_kernel<2>", "rows_kernel<2", "rows256_kernel<sr::BabyBear, 2", "rows_kernel<sr::BabyBear, 2", "tile_kernel<", "rows_kernel<sr::")
                 if args.variant == "mul" else ("rows256_kernel<3>", "rows_kernel<3"),
                 "fwd_cols": ("cols256_keep_kernel<0>", "cols256_pair_kernel", "cols256_kernel<0", "cols256_kernel<sr::BabyBear, 0", "cols_kernel<3, 0", "cols_kernel<2, 0", "cols_kernel<1, 0", "strided"),
                 "inv_cols": ("cols256_keep_kernel<1>", "cols256_kernel<1", "cols256_kernel<sr::BabyBear, 1", "cols_kernel<3, 1", "cols_kernel<2, 1", "cols_kernel<1, 1", "strided")}.get(dom_tag, ())
    rocprof_ms, rocprof_kernel, rocprof_source = replayed_kernel_stats(args.workload, dom_names, source_hash()) if dom_names else (None, None, "n/a")
    if rocprof_ms and "dropped (other plan)" in traffic_source:
        rocprof_ms, rocprof_source = None, "dropped (other plan): the profile was collected with other launches per step"
    frac_events = achieved_gbs / HBM_PEAK_GBS
    frac_rocprof = (alg_bytes_per_launch / (rocprof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rocprof_ms else None
    frac_reported = min(frac_events, frac_rocprof) if frac_rocprof else frac_events
    # (b) whole-step VALU issue: wave-instructions of the step / step time, against one wave-instruction per 4 cycles per SIMD (256 CUs x
    #     4 SIMDs) at the clock the chip HELD during the timed steps (sampled) and at the nominal 2.4 GHz
    step_s = elapsed / args.steps
    whole_valu = None
    if valu_step:
        rate = valu_step / step_s
        sclk = clock_power["sclk_mhz"] * 1e6 if clock_power else None
        whole_valu = {"valu_wave_instructions_per_step": valu_step, "achieved": rate, "unit": "wave-instructions/s",
                      "peak_at_sampled_clock": (1024 * sclk / 4) if sclk else None, "frac": (rate / (1024 * sclk / 4)) if sclk else None,
                      "peak_at_nominal_2400_mhz": 1024 * 2.4e9 / 4, "frac_at_nominal_2400_mhz": rate / (1024 * 2.4e9 / 4),
                      "issue_floor_ms_at_sampled_clock": (valu_step / (1024 * sclk / 4) * 1e3) if sclk else None,
                      "sclk_mhz": clock_power["sclk_mhz"] if clock_power else None, "socket_w": clock_power["socket_w"] if clock_power else None,
                      "scaled": valu_scaled,
                      "note": "VALU per wave and waves per launch replayed from the PMC pass under profiles/ (hash-guarded like roofline.traffic); "
                              "launches per step, step time, clock and power measured in this run"}
    hbm_traffic_frac = (step_traffic / step_s / 1e9 / HBM_PEAK_GBS) if step_traffic else None
    vfrac = whole_valu and (whole_valu["frac"] or whole_valu["frac_at_nominal_2400_mhz"])
    if vfrac and vfrac > 0.75 and hbm_traffic_frac is not None and hbm_traffic_frac < 0.8:
        limiter = "valu-issue @ power cap"
    elif hbm_traffic_frac is not None and hbm_traffic_frac >= 0.8:
        limiter = "hbm"
    elif vfrac:
        limiter = "valu-issue (%.2f of the issue rate at the %s clock) + latency / launch tails" % (vfrac, "sampled" if whole_valu["frac"] else "nominal")
    else:
        limiter = "undetermined: no PMC replay for this build (profiles/ carries another source hash)"

    out = {
        "metric": ("ring-muls/sec (Goldilocks, deg 2^16, batch 2^14)" if args.workload == "goldilocks_d65536_b16384"
                   else "ring-muls/sec (%s)" % args.workload) + ("" if args.variant == "mul" else " [variant %s: b in NTT form]" % args.variant),
        "value": value,
        "unit": "ring-muls/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"goldilocks": "u64", "babybear": "u32", "stark": "u256"}[ring_name],
        "data": "synthetic",
        "config": {"workload": args.workload, "ring": ring_name, "degree": d, "batch_per_gpu": batch,
                   "global_batch": batch * world,
                   "layout": ("packed uint32 = low half of the ark-ff Fp64 Montgomery limb (opt-in sr_*_packed32_* entry points), in place (a *= b), b read-only"
                              if packed else "ark-ff Montgomery u64 limbs, in place (a *= b), b read-only"),
                   "variant": args.variant, "plan": plan_used,
                   "parallelism": "batch-sharded x%d, twiddle broadcast only" % world},
        "roofline": {"bound": "hbm", "achieved": frac_reported * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": frac_reported, "traffic": traffic, "traffic_source": traffic_source,
                     "limiter": limiter,
                     "frac_is": "the lower of frac_events and frac_rocprof (the same algorithmic bytes per launch over two measurements of the launch's duration)",
                     "frac_events": frac_events, "frac_rocprof": frac_rocprof,
                     "kernel_avg_ms_events": dom_avg_ms, "kernel_avg_ms_rocprof": rocprof_ms, "rocprof_kernel": rocprof_kernel, "rocprof_source": rocprof_source,
                     "events": {"bracketed": "every %d%s launch" % (PROF_STRIDE, "th" if PROF_STRIDE > 1 else "st"), "sampled_launches": dom.get("sampled"),
                                "profiled_step_ms": profile_step_ms, "timed_step_ms": elapsed / args.steps * 1e3,
                                "lane_time_ms_per_step": in_flight,
                                "note": "HIP events on the launch stream around a sparse sample of the launches over K more steps; profiled_step_ms ~ "
                                        "timed_step_ms says the sampling left the step as it runs"},
                     "copy_measured": copy_gbs, "copy_measured_kind": copy_kind,
                     "frac_of_copy_measured": frac_reported * HBM_PEAK_GBS / copy_gbs if copy_gbs else None,
                     "kernel": dom_tag, "kernel_avg_ms": (alg_bytes_per_launch / (frac_reported * HBM_PEAK_GBS * 1e9) * 1e3), "launches_per_step": launches_per_step, "by_kernel": by_kernel,
                     "launches_per_step_by_kernel": {t: v["launches"] / args.steps for t, v in kern.items()},
                     "algorithmic_bytes_per_ring_mul": bytes_per_mul,
                     "dominant_kernel_bytes_per_ring_mul": kernel_bytes_per_elem,
                     "whole_step_achieved_per_gpu": step_gbs, "whole_step_frac": step_gbs / HBM_PEAK_GBS,
                     # measured HBM traffic of the whole step (PMC passes under profiles/) over the step time: the bandwidth the
                     # step actually draws, as opposed to the algorithmic 3*D*w figure above
                     "whole_step_hbm_traffic_bytes": step_traffic,
                     "whole_step_hbm_traffic_gbs": (step_traffic / (elapsed / args.steps) / 1e9) if step_traffic else None,
                     "whole_step_hbm_traffic_frac": (step_traffic / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
                     "per_kernel_ms_per_step": {t: v["ms"] / args.steps for t, v in kern.items()}},
    }
    if prof_single:
        # what `kernel_avg_ms` above means when launches of two streams overlap, and the one-stream figures next to it
        ps = prof_single[dom_tag]
        s_launch_bytes = kernel_bytes_per_elem * batch / (ps["launches"] / ps["steps"]) * (fwd_operands if dom_tag == "fwd_cols" else 1)
        s_avg = ps["ms"] / ps["launches"]
        out["roofline"]["overlap"] = {
            "internal_streams": 2, "kernel_ms_in_flight_per_step": in_flight, "factor": in_flight / (elapsed / args.steps * 1e3),
            "note": "chunks of the batch run on two internal streams: a launch's event-timed duration is time in flight beside the other "
                    "stream's kernels, so roofline.achieved (algorithmic bytes of a launch / that duration, as specified) is lower than "
                    "the same kernel alone; whole_step_* is the figure unaffected by the overlap"}
        out["roofline"]["single_stream"] = {
            "what": "calibration outside the timed region: the same product on a second context with sr_plan.lanes = 1 (one stream, eight "
                    "large chunks) -- NOT the configuration `value` was measured on",
            "ms_per_step": single_ms, "kernel": dom_tag, "kernel_avg_ms": s_avg, "launches_per_step": ps["launches"] / ps["steps"],
            "achieved": s_launch_bytes / (s_avg * 1e-3) / 1e9, "frac": s_launch_bytes / (s_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "per_kernel_ms_per_step": {t: v["ms"] / v["steps"] for t, v in prof_single.items()}}

    if valu is not None or whole_valu is not None:
        out["integer_valu"] = dict(valu or {}, bound=limiter if "valu" in limiter else (valu or {}).get("bound", "valu-issue"), whole_step=whole_valu)
    if clock_power:
        out["clock_power"] = clock_power
    if parity:
        out["parity"] = parity
    if dist_on:
        out["multi_gpu"] = {"backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""), "ranks_seen": ranks_seen,
                            "devices_visible": ndev, "twiddle_broadcast_bytes": bcast_bytes, "twiddle_broadcast_ms": bcast_ms,
                            "data_path_collectives": 0, "rank_ms_per_step": rank_ms,
                            "collectives_run": ["broadcast(twiddle block, device view)", "all_reduce(rank count)",
                                                "all_reduce(step time: max, min)"] + (["all_reduce(parity: min)"] if parity else [])}

    if world == 1 and not args.no_cpu_baseline:
        import oracle_lib as O  # the timed CPU baseline (the parity check above used it as the checker)

        cflags = O.use_native_build()  # the timed build is -O3 -march=native, compiled on this box for this box
        F = O.FIELD_ID[ring_name]
        cores = usable_cores()
        n0 = max(cores, 8)
        ea = O.fill_uniform(F, 1, 0, n0 * d)
        eb = O.fill_uniform(F, 2, 0, n0 * d)
        t1 = time.perf_counter()
        O.pow2_ring_mul(F, ea, eb, k, n0, cores)
        probe = time.perf_counter() - t1
        n = int(min(max(n0, n0 * args.cpu_seconds / max(probe, 1e-6)), 1 << 15))
        n = max(cores, (n // cores) * cores)
        ea = O.fill_uniform(F, 1, 0, n * d)
        eb = O.fill_uniform(F, 2, 0, n * d)
        t1 = time.perf_counter()
        O.pow2_ring_mul(F, ea, eb, k, n, cores)
        cpu_t = time.perf_counter() - t1
        t1 = time.perf_counter()
        n1 = max(1, n // cores // 4)
        O.pow2_ring_mul(F, ea[:n1 * ring.words_per_elem], eb[:n1 * ring.words_per_elem], k, n1, 1)
        cpu_t1 = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": n / cpu_t, "unit": "ring-muls/s", "cores": cores, "kind": "port",
            "sample": "%d ring-muls of the same workload (D=2^%d), %d pthreads over the batch, %.1f s; "
                      "C restatement of the reference CPU path (oracle/sr_oracle.c, gcc %s, built on this box), not the Rust binary"
                      % (n, k, cores, cpu_t, cflags),
            "single_thread_value": n1 / cpu_t1,
        }
        out["gpu_over_cpu"] = value / (n / cpu_t)
        # what `RqPoly * RqPoly` really runs in the reference (coeff_form.rs:54-67): O(D^2) schoolbook + reduce, one thread,
        # timed on one element at D' = min(D, 4096) and scaled by (D / D')^2 -- reported separately, never the headline
        ds = min(d, 4096)
        t1 = time.perf_counter()
        O.schoolbook(F, ea[:ds * O.LIMBS[F]], eb[:ds * O.LIMBS[F]], ds)
        sb = time.perf_counter() - t1
        out["cpu_baseline"]["schoolbook"] = {"degree": ds, "seconds_per_product": sb,
                                             "ring_muls_per_s_scaled_to_workload_degree": 1.0 / (sb * (d / ds) ** 2)}
    print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
