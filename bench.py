#!/usr/bin/env python3
"""Headline benchmark: batched ring multiplications c = a * b in Fp[X]/(X^D+1) on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One "step" = one pass of the hot path (forward NTT x2, slot product, inverse NTT) over one batch of
synthetic coefficient vectors already resident in HBM.  Default workload = BASELINE.json configs[1]:
Goldilocks, D = 2^16, batch = 2^14 per GPU (weak scaling: the batch shards by element, no data-path
collective; the only collective is one RCCL broadcast of the twiddle block at start-up).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel, HIP
events on the launch stream) and `cpu_baseline` (the oracle -- a C restatement of the reference's CPU
path -- timed on this host's cores over a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    # name: (ring, log2 D, batch per GPU, bytes per coefficient)
    "goldilocks_d65536_b16384": ("goldilocks", 16, 1 << 14, 8),   # BASELINE configs[1] (metric config)
    "babybear_d65536_b16384": ("babybear", 16, 1 << 14, 8),       # configs[2], reference 8-byte layout
    "goldilocks_d1048576_b8192": ("goldilocks", 20, 1 << 13, 8),  # configs[3] per-GPU shard (2^16 / 8)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="goldilocks_d65536_b16384", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

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
    dev_index = local_rank % ndev  # one rank per GPU on a full node; a 1-GPU rehearsal folds ranks onto device 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    ring_name, k, batch, coeff_bytes = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    d = 1 << k
    ring = CyclotomicRing(ring_name, k, device=dev_index)
    words = batch * ring.words_per_elem

    # ---- shared twiddles: rank 0's tables broadcast once over RCCL/xGMI, adopted by every rank ----
    if world > 1:
        from stark_rings_amd.sharding import share_twiddles

        share_twiddles(ring, dev)  # non-zero ranks zero their block first: the tables really come from rank 0

    # ---- synthetic inputs generated on device (counter-based PRNG; rank-disjoint coefficient ranges) ----
    a = torch.empty(words, dtype=torch.int64, device=dev)
    b = torch.empty(words, dtype=torch.int64, device=dev)
    first = rank * batch * d
    ring.fill_uniform_dev(a, 0x5EED0001, first)
    ring.fill_uniform_dev(b, 0x5EED0002, first)
    torch.cuda.synchronize()

    # ---- property gate (no oracle here: the oracle is only used inside the cpu_baseline leg below): one untimed step,
    #      outputs canonical, icrt(crt(c)) == c on the result, sampled outputs kept for the cpu_baseline leg's bit-exact check
    wpe = ring.words_per_elem
    ring.mul_dev(a, a, b)
    torch.cuda.synchronize()
    assert ring.count_noncanonical_dev(a) == 0, "non-canonical outputs"
    sample = sorted({0, batch // 3, batch - 1})
    sample_out = {e: a[e * wpe:(e + 1) * wpe].cpu().numpy().view(np.uint64).copy() for e in sample}
    rt = a[:min(batch, 8) * wpe].clone()
    ring.elementwise_crt_dev(rt)
    ring.elementwise_icrt_dev(rt)
    if not torch.equal(rt, a[:rt.numel()]):
        raise SystemExit("PROPERTY FAILURE rank %d: icrt(crt(c)) != c" % rank)
    del rt

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ring.mul_dev(a, a, b)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ring.mul_dev(a, a, b)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations over the same K steps, HIP events on the launch stream ----
    ring.profile_enable(True)
    for _ in range(args.steps):
        ring.mul_dev(a, a, b)
    torch.cuda.synchronize()
    prof = ring.profile_read()
    ring.profile_enable(False)

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

        nb = a.numel() * 8
        rates = {"hipMemcpy device-to-device (1 read + 1 write stream)": stream_rate(lambda: b.copy_(a), 2 * nb),
                 "element-wise ring add a += b (2 read + 1 write streams)": stream_rate(lambda: ring.add_dev(b, a), 3 * nb)}
        copy_kind = max(rates, key=rates.get)
        copy_gbs = rates[copy_kind]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_muls = world * batch * args.steps
    value = total_muls / elapsed
    bytes_per_mul = 3 * d * coeff_bytes  # read a, read b, write c
    kern = {t: v for t, v in prof.items() if v["launches"]}
    # dominant kernel = the longest single launch of the step (the two forward column launches together take about as long as
    # the rows launch at D = 2^16, so "largest total" would flip between runs; the per-launch figure is stable)
    dom_tag = max(kern, key=lambda t: kern[t]["ms"] / kern[t]["launches"])
    dom = kern[dom_tag]
    launches_per_step = dom["launches"] / args.steps
    dom_avg_ms = dom["ms"] / dom["launches"]
    # algorithmic bytes one launch of the dominant kernel is responsible for: the whole batch's 3*D*w,
    # split over the launches of that kernel in one step
    alg_bytes_per_launch = bytes_per_mul * batch / launches_per_step
    achieved_gbs = alg_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9
    step_gbs = bytes_per_mul * batch * world / (elapsed / args.steps) / 1e9 / world

    # HBM traffic of the dominant kernel per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in their own runs, FETCH_SIZE doubled per MI355X_MICROARCH.md): profiles/<round>/traffic.json
    traffic = None
    step_traffic = None  # measured HBM bytes of one whole step (every launch), for the achieved-bandwidth figure
    try:
        rounds = sorted(r for r in os.listdir(os.path.join(ROOT, "profiles")) if r.startswith("r"))
        tj = json.load(open(os.path.join(ROOT, "profiles", rounds[-1], "traffic.json")))
        if tj.get("workload") == args.workload and batch == tj.get("batch"):
            traffic = tj["bytes_per_launch"].get(dom_tag)
            if all(t in tj["bytes_per_launch"] for t in kern):
                step_traffic = sum(tj["bytes_per_launch"][t] * kern[t]["launches"] / args.steps for t in kern)
    except (OSError, ValueError, KeyError, IndexError):
        pass

    # integer-VALU issue roofline of the same kernel: dynamic VALU instructions per wave (SQ counter pass committed
    # under profiles/) x waves per launch / measured duration, against one wave-instruction per 4 cycles per SIMD
    valu = None
    try:
        vpath = os.path.join(ROOT, "profiles", rounds[-1], "valu_%s.json" % args.workload)
        if not os.path.exists(vpath):
            vpath = os.path.join(ROOT, "profiles", rounds[-1], "valu.json")
        vj = json.load(open(vpath))
        if vj.get("workload") == args.workload and batch == vj.get("batch") and dom_tag in vj["kernels"]:
            kv = vj["kernels"][dom_tag]
            rate = kv["waves_per_launch"] * kv["valu_per_wave"] / (dom_avg_ms * 1e-3)
            peak = 256 * 4 * 2.4e9 / 4
            valu = {"bound": "valu-issue", "achieved": rate, "peak": peak, "unit": "wave-instructions/s", "frac": rate / peak,
                    "valu_instructions_per_wave": kv["valu_per_wave"], "waves_per_launch": kv["waves_per_launch"],
                    "note": "supplementary: the path is integer-VALU-bound, see DESIGN.md section 6"}
    except (OSError, ValueError, KeyError, IndexError, NameError):
        pass

    out = {
        "metric": "ring-muls/sec (Goldilocks, deg 2^16, batch 2^14)" if args.workload == "goldilocks_d65536_b16384"
                  else "ring-muls/sec (%s)" % args.workload,
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
                   "global_batch": batch * world, "layout": "ark-ff Montgomery u64 limbs, in place (a *= b)",
                   "parallelism": "batch-sharded x%d, twiddle broadcast only" % world},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "copy_measured": copy_gbs, "copy_measured_kind": copy_kind,
                     "frac_of_copy_measured": achieved_gbs / copy_gbs if copy_gbs else None,
                     "kernel": dom_tag, "kernel_avg_ms": dom_avg_ms, "launches_per_step": launches_per_step,
                     "algorithmic_bytes_per_ring_mul": bytes_per_mul,
                     "whole_step_achieved_per_gpu": step_gbs, "whole_step_frac": step_gbs / HBM_PEAK_GBS,
                     # measured HBM traffic of the whole step (PMC passes under profiles/) over the step time: the bandwidth the
                     # step actually draws, as opposed to the algorithmic 3*D*w figure above
                     "whole_step_hbm_traffic_bytes": step_traffic,
                     "whole_step_hbm_traffic_gbs": (step_traffic / (elapsed / args.steps) / 1e9) if step_traffic else None,
                     "whole_step_hbm_traffic_frac": (step_traffic / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
                     "per_kernel_ms_per_step": {t: v["ms"] / args.steps for t, v in kern.items()}},
    }

    if valu is not None:
        out["integer_valu"] = valu

    if world == 1 and not args.no_cpu_baseline:
        import oracle_lib as O  # test infrastructure; used here as CHECKER of the sampled GPU outputs and as the timed CPU baseline

        F = O.FIELD_ID[ring_name]
        for e in sample:
            ea = O.fill_uniform(F, 0x5EED0001, first + e * d, d)
            eb = O.fill_uniform(F, 0x5EED0002, first + e * d, d)
            if not np.array_equal(sample_out[e], O.pow2_ring_mul(F, ea, eb, k)):
                raise SystemExit("PARITY FAILURE element %d: GPU result differs from the oracle" % e)
        out["parity"] = "bit-exact vs oracle on elements %s of the first step; all outputs canonical; icrt(crt(c)) == c" % sample
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
                      "C restatement of the reference CPU path (oracle/sr_oracle.c), not the Rust binary" % (n, k, cores, cpu_t),
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
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
