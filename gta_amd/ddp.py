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
    and keeps a timeline: per bucket its bytes and -- on CUDA -- two events: one recorded on the compute stream when the
    bucket is handed over, one in the future's callback when the reduced bucket has been scaled.  Their distance is the
    bucket's ISSUE-TO-RESULT latency: it contains the wait for backward kernels queued in front of the collective and the
    callback's own latency, not the collective alone (RCCL's own kernels show up in a rocprofv3 kernel trace).  The Python hook
    also replaces DDP's built-in C++ all-reduce, so a step timed with it is slightly perturbed: bench.py times the
    optimizer steps WITHOUT the hook and logs buckets on a few extra steps."""

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
                "bucket_issue_to_result_ms_per_step": (sum(ms) / max(steps, 1)) if ms else None,
                "note": "per bucket: handed to the hook on the compute stream -> reduced and scaled (includes queued backward kernels "
                        "and callback latency; not the collective alone); buckets overlap the backward"}


def wrap_ddp_logged(model: torch.nn.Module, local_rank: int = 0, bucket_mb: int = XGMI_BUCKET_MB, force: bool = False):
    """wrap_ddp + a BucketLog registered as the communication hook: (model, log); (model, None) when single process
    (``force``: wrap also in a world of one -- the reducer, its bucket views and the hook then run for real on one rank)."""
    if force and dist.is_available() and dist.is_initialized() and world()[1] == 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        cuda = next(model.parameters()).is_cuda
        m = DDP(model, device_ids=[local_rank] if cuda else None, bucket_cap_mb=bucket_mb, gradient_as_bucket_view=cuda)
    else:
        m = wrap_ddp(model, local_rank, bucket_mb)
    if m is model:
        return m, None
    log = BucketLog()
    m.register_comm_hook(None, log.hook)
    return m, log


def wrap_srt_ddp(model: torch.nn.Module, local_rank: int = 0, logged: bool = False, bucket_mb: int = XGMI_BUCKET_MB,
                 force: bool = False):
    """The reference's data-parallel structure (train.py:182-188): ``model.encoder`` and ``model.decoder`` each in their OWN
    DistributedDataParallel -- two reducers, two bucket streams.  Returns (model, [logs]) with the sub-modules replaced in place;
    the logs are BucketLog hooks when ``logged`` (else empty)."""
    logs = []
    for name in ("encoder", "decoder"):
        sub = getattr(model, name)
        if logged:
            w, log = wrap_ddp_logged(sub, local_rank, bucket_mb, force=force)
            if log is not None:
                logs.append(log)
        else:
            w = wrap_ddp(sub, local_rank, bucket_mb)
        setattr(model, name, w)
    return model, logs


def backend_info() -> dict:
    """what the process group runs on, for the bench line: backend, world size, RCCL version (torch's nccl IS RCCL on ROCm)"""
    info = {"backend": None, "world": 1, "rccl_version": None}
    if dist.is_available() and dist.is_initialized():
        info["backend"] = dist.get_backend()
        info["world"] = dist.get_world_size()
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        pass
    return info


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX of a host-measured duration over ranks (bench.py contract)."""
    r, w = world()
    if w == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
