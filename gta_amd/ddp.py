"""Data-parallel harness -- one process per GPU, RCCL over xGMI (``backend='nccl'`` on ROCm).

Mirrors the reference's distributed plumbing: ``init_ddp`` (source/utils/common.py:18-30),
``gather_all`` (:69-77), ``reduce_dict`` (:80-102), per-rank batch = global batch // world size
(train.py:110) and DDP over the model (train.py:182-188).  The GTA operator itself has no
exchange step (every (batch, head, query tile) is independent), so the only data-path collective
is DDP's bucketed gradient all-reduce, overlapped with backward.

xGMI is point-to-point (7 links x ~153 GB/s per GPU); the whole gradient is 58.8 MB (CLEVR-TR gta)
or 163 MB (MSN gta_so3) of fp32, so a few large buckets keep every link busy: the default bucket
is 64 MB instead of DDP's NVSwitch-era 25 MB.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch
import torch.distributed as dist

XGMI_BUCKET_MB = 64


def init_ddp(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; single process when absent."""
    if "WORLD_SIZE" not in os.environ or int(os.environ["WORLD_SIZE"]) <= 1:
        return 0, 1, 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this platform
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return rank, world, local_rank


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world_size: int) -> Dict[str, torch.Tensor]:
    """Contiguous slice of the global batch for this rank (reference: batch_size // world_size per
    process with a DistributedSampler, train.py:110,141-145).  The global batch must divide evenly."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        if n % world_size:
            raise ValueError(f"global batch {n} of '{k}' does not divide over {world_size} ranks")
        per = n // world_size
        out[k] = v[rank * per:(rank + 1) * per]
    return out


def gather_all(t: torch.Tensor) -> torch.Tensor:
    """Concatenate a per-rank tensor over ranks (common.py:69-77)."""
    r, w = world()
    if w == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(w)]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, 0)


def reduce_dict(d: Dict[str, torch.Tensor], average: bool = True) -> Dict[str, torch.Tensor]:
    """All-reduce every entry in key order (common.py:80-102); one fused call instead of one per key."""
    r, w = world()
    if w == 1:
        return d
    keys = sorted(d)
    flat = torch.cat([d[k].reshape(-1).float() for k in keys])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= w
    out, off = {}, 0
    for k in keys:
        n = d[k].numel()
        out[k] = flat[off:off + n].reshape(d[k].shape).to(d[k].dtype)
        off += n
    return out


def wrap_ddp(model: torch.nn.Module, local_rank: int = 0, bucket_mb: int = XGMI_BUCKET_MB) -> torch.nn.Module:
    """DistributedDataParallel with xGMI-sized buckets (no-op when single process)."""
    r, w = world()
    if w == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    if next(model.parameters()).is_cuda:
        return DDP(model, device_ids=[local_rank], bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True)
    return DDP(model, bucket_cap_mb=bucket_mb)


class BucketLog:
    """DDP communication hook that does what the default hook does (all-reduce the bucket, divide by the world size)
    and keeps a timeline: per bucket its bytes and -- on CUDA -- events around the collective on the stream it runs on.
    ``summary()`` after a synchronise gives bytes, bucket count and the summed all-reduce time of the logged steps, so
    ``bench.py --model-train-steps`` can report the collective beside the step."""

    def __init__(self):
        self.rows = []

    def reset(self):
        self.rows = []

    def hook(self, state, bucket):
        t = bucket.buffer()
        w = dist.get_world_size()
        ev = None
        if t.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        fut = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True).get_future()
        row = {"bytes": t.numel() * t.element_size(), "events": ev}
        self.rows.append(row)

        def done(f):
            out = f.value()[0]
            out.div_(w)
            if ev is not None:
                ev[1].record()
            return out
        return fut.then(done)

    def summary(self, steps: int = 1):
        if not self.rows:
            return None
        ms = [r["events"][0].elapsed_time(r["events"][1]) for r in self.rows if r["events"] is not None]
        return {"buckets_per_step": len(self.rows) / max(steps, 1), "bytes_per_step": sum(r["bytes"] for r in self.rows) / max(steps, 1),
                "allreduce_ms_per_step": (sum(ms) / max(steps, 1)) if ms else None,
                "note": "events around each bucket's all-reduce (issue -> result scaled); buckets overlap the backward"}


def wrap_ddp_logged(model: torch.nn.Module, local_rank: int = 0, bucket_mb: int = XGMI_BUCKET_MB):
    """wrap_ddp + a BucketLog registered as the communication hook: (model, log); (model, None) when single process."""
    m = wrap_ddp(model, local_rank, bucket_mb)
    if m is model:
        return m, None
    log = BucketLog()
    m.register_comm_hook(None, log.hook)
    return m, log


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX of a host-measured duration over ranks (bench.py contract)."""
    r, w = world()
    if w == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
