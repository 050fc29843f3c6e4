"""Batch sharding across the GPUs of one node: one process per GPU, torch.distributed (RCCL = "nccl").

The hot path partitions by ring element: element e of a batch touches only a[e], b[e], c[e]
(SURVEY.md 8e), so ranks own contiguous element ranges and exchange NOTHING in steady state.  The only
collective is a one-shot broadcast of rank 0's twiddle block at start-up (north_star: "RCCL broadcast of
shared twiddles over xGMI only").  The same code runs over gloo on CPU tensors, which is how the
multi-rank logic is tested without GPUs (tests/test_sharding_gloo.py).
"""
import os


def shard_range(batch, world, rank):
    """Contiguous, balanced split of `batch` elements: returns (first, count) for `rank`.
    The first batch % world ranks get one extra element."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    q, r = divmod(batch, world)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def group_device_ids(device_count, cap=8):
    """Device ids for a single-process context group (sr_ctx_create_group) on a box with `device_count` visible GPUs: every visible
    device up to `cap`, DISTINCT ids (the peer copy of the twiddle block then really crosses xGMI); with one GPU -- or none visible --
    two contexts on device 0, the one-GPU form of the same call path."""
    n = min(int(device_count), int(cap))
    return list(range(n)) if n >= 2 else [0, 0]


def rehearsal_backend(device_count):
    """(backend, ranks) for the self-launched multi-rank rehearsal of bench.py on this box: RCCL ("nccl") over distinct devices as
    soon as there are two -- two ranks, a real inter-GPU broadcast -- otherwise two gloo ranks folded onto the one GPU."""
    return ("nccl", 2) if int(device_count) >= 2 else ("gloo", 2)


def env_world():
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) when run stand-alone."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend, rank, world, device=None):
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    kwargs = {}
    if device is not None and backend == "nccl":
        kwargs["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return dist


class DeviceBytes:
    """Zero-copy torch view (uint8) of device memory owned by the C library (__cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_block(block, src=0):
    """Broadcast a flat byte tensor (CPU for gloo, CUDA for nccl/RCCL) from `src` to every rank, in place.
    Non-source ranks are zeroed first so that a stale local table can never masquerade as the broadcast."""
    import torch.distributed as dist

    if dist.get_rank() != src:
        block.zero_()
    dist.broadcast(block, src=src)
    return block


def share_twiddles(ring, device):
    """GPU path: rank 0's twiddle block (generic + tuned tables, one allocation) -> every rank's context."""
    import torch

    ptr, nbytes = ring.twiddle_block()
    view = torch.as_tensor(DeviceBytes(ptr, nbytes), device=device)
    broadcast_block(view, src=0)
    torch.cuda.synchronize()
    ring.twiddles_updated()
    return nbytes
