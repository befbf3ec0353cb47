"""View-sharded data parallelism for the SDS loop: one process per GPU, RCCL over xGMI.

The reference is single-GPU only (Garment_3DGS/generate_3dgs.py:40,57-58 ``devices=1``); its
implicit coupling between views is autograd accumulating ``.grad`` over the Python view loop
(Garment_3DGS/threestudio/systems/GaussianDreamer.py:189-191) plus two batch-wide reductions:
``depths.max()`` in the sparsity head (:215) and the per-view ``viewspace`` gradient sum /
``radii`` max used by densification (:268-279).  Sharding the V views over N ranks therefore needs
exactly:
  * ONE all-reduce(sum) per iteration over a single flat fp32 buffer
    [all parameter grads | summed viewspace grads], scaled by 1/N so the result equals the
    single-GPU gradient (``loss_sds`` is normalised by the LOCAL batch size,
    stable_diffusion_guidance.py:427, and every rank holds V/N views);
  * ONE all-reduce(max) of an int32 buffer [``radii`` | bits of this rank's ``depths.max()``], started
    asynchronously right after the rasterizer's forward pass and waited on where its first consumer (the sparsity
    head, after the guidance forward) runs: it overlaps the VAE / UNet work (round 4; rounds 1-3 ran a blocking
    scalar max after the render and a blocking radii max after the backward pass);
  * a scalar sum in the backward pass of the depth maximum (the gradient goes to the rank that owns it).
xGMI is point-to-point, so per-iteration traffic is kept to one ~7 MB bucket (P = 100k) instead
of one collective per parameter tensor.  Parameters and optimizer state are replicated.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def collectives_on() -> bool:
    """True when the loop must run its collectives: a group of more than one rank, or the one-rank group of
    ``GD_DIST_SINGLE=1`` (hardware check of the RCCL path of every collective on a single-GPU box: with one rank each
    of them is an identity, but it is issued on the "nccl" backend with the loop's tensors, dtypes and streams)."""
    return is_dist() and (dist.get_world_size() > 1 or os.environ.get("GD_DIST_SINGLE") == "1")


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; no-op for 1 process.
    Returns (rank, local_rank, world_size)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    # GD_DIST_SINGLE=1: a process group of ONE rank (tests: the RCCL code path of every collective on a single-GPU box)
    if (ws > 1 or os.environ.get("GD_DIST_SINGLE") == "1") and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # GD_DIST_BACKEND=gloo lets several ranks share ONE GPU (functional check of the sharded
            # path on a single-GPU box); production is "nccl" (= RCCL on ROCm), one rank per GPU.
            backend = os.environ.get("GD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(lr)
            dist.init_process_group(backend, rank=rk, world_size=ws, device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend, rank=rk, world_size=ws)
    return rk, lr, ws


def shard_views(n_views: int, rk: Optional[int] = None, ws: Optional[int] = None) -> List[int]:
    """Rank r renders views r, r+N, r+2N, ...  (requires n_views % N == 0 so every rank's local
    batch has the same size -- the 1/N gradient scaling depends on it)."""
    rk = rank() if rk is None else rk
    ws = world_size() if ws is None else ws
    if n_views % ws != 0:
        raise ValueError(f"n_views={n_views} must be divisible by world size {ws}")
    return list(range(rk, n_views, ws))


class _GlobalMax(torch.autograd.Function):
    """max over all ranks of a per-rank scalar, differentiable like ``Tensor.max()``: the gradient
    (summed over ranks) flows to the rank that holds the maximum."""

    @staticmethod
    def forward(ctx, local_max: torch.Tensor) -> torch.Tensor:
        g = local_max.detach().clone()
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        ctx.save_for_backward(local_max.detach() >= g)
        return g

    @staticmethod
    def backward(ctx, grad_out):
        (is_owner,) = ctx.saved_tensors
        g = grad_out.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return torch.where(is_owner, g, torch.zeros_like(g))


def global_max(local_max: torch.Tensor) -> torch.Tensor:
    return _GlobalMax.apply(local_max) if collectives_on() else local_max


class _GlobalMaxKnown(torch.autograd.Function):
    """``_GlobalMax`` whose forward collective has already run (``PendingMax``): only the backward sum is left."""

    @staticmethod
    def forward(ctx, local_max: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(local_max.detach() >= g)
        return g.clone()

    @staticmethod
    def backward(ctx, grad_out):
        (is_owner,) = ctx.saved_tensors
        g = grad_out.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return torch.where(is_owner, g, torch.zeros_like(g)), None


class PendingMax:
    """One asynchronous all-reduce(max) carrying the per-Gaussian ``radii`` (int32, max over this rank's views) AND
    this rank's ``depths.max()``: a non-negative fp32 value orders like its bit pattern read as int32, so the scalar
    rides in the tail slot of the integer buffer.  ``start`` right after the rasterizer's forward pass, ``finish``
    where the results are first needed -- the collective then runs beside the guidance forward on RCCL's stream."""

    def __init__(self, radii_local: torch.Tensor, local_max: torch.Tensor, flag: Optional[torch.Tensor] = None):
        """``flag``: optional int32 [1] device tensor that rides in one more tail slot (max over ranks = logical OR of a 0/1
        flag): the sync-free rasterizer's overflow flag, so that every rank takes the same repeat-the-render decision
        (SDSLoop.step).  Its reduced value travels to pinned host memory on a side stream as soon as the collective ends;
        ``flag_any()`` waits for that copy only -- not for the work queued on the caller's stream in the meantime."""
        if radii_local.dtype != torch.int32 or local_max.dtype != torch.float32:
            raise TypeError("PendingMax: radii must be int32 and the depth maximum fp32")
        self.local_max = local_max
        self.n = radii_local.numel()
        # clamp: a negative zero / negative value would order wrongly as an integer (depths are sums of w * depth >= 0)
        tail = local_max.detach().clamp_min(0.0).reshape(1).view(torch.int32)
        parts = [radii_local.reshape(-1), tail]
        self._has_flag = flag is not None
        if self._has_flag:
            parts.append(flag.reshape(1).to(torch.int32))
        self.buf = torch.cat(parts)
        self.work = dist.all_reduce(self.buf, op=dist.ReduceOp.MAX, async_op=True) if collectives_on() else None
        self._flag_host = self._flag_event = None
        if self._has_flag:
            if self.buf.is_cuda:
                cur = torch.cuda.current_stream(self.buf.device)
                side = _flag_stream(self.buf.device)
                side.wait_stream(cur)                     # the buffer was assembled on the caller's stream
                with torch.cuda.stream(side):
                    if self.work is not None:
                        self.work.wait()                  # RCCL: orders `side` behind the collective, no host wait
                    self._flag_host = torch.empty(1, dtype=torch.int32).pin_memory()
                    self._flag_host.copy_(self.buf[self.n + 1:], non_blocking=True)
                    self._flag_event = torch.cuda.Event()
                    self._flag_event.record(side)
                self.buf.record_stream(side)
            else:
                if self.work is not None:
                    self.work.wait()
                    self.work = None
                self._flag_host = self.buf[self.n + 1:].clone()

    def flag_any(self) -> bool:
        """Whether ANY rank raised the flag (identical on every rank)."""
        if not self._has_flag:
            return False
        if self._flag_event is not None:
            self._flag_event.synchronize()
        return bool(int(self._flag_host[0]))

    def finish(self):
        """-> (radii max over every rank's views [P] int32, global depth maximum attached to the autograd graph)."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        g = self.buf[self.n:self.n + 1].view(torch.float32).reshape(())
        radii = self.buf[:self.n]
        if collectives_on():
            return radii, _GlobalMaxKnown.apply(self.local_max, g)
        return radii, self.local_max


_FLAG_STREAMS = {}


def _flag_stream(device):
    """One side stream per device for the reduced overflow flag's trip to the host (PendingMax)."""
    key = torch.device(device).index
    if key not in _FLAG_STREAMS:
        _FLAG_STREAMS[key] = torch.cuda.Stream(device)
    return _FLAG_STREAMS[key]


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def scale_grad(x: torch.Tensor, s: float) -> torch.Tensor:
    """Identity in forward, multiplies the gradient by ``s`` in backward."""
    return _ScaleGrad.apply(x, s)


class GradBucket:
    """Flat fp32 bucket: copy grads in, one all-reduce, scale, copy back."""

    def __init__(self, tensors: Sequence[torch.Tensor]):
        self.shapes = [t.shape for t in tensors]
        self.numels = [t.numel() for t in tensors]
        dev = tensors[0].device
        self.flat = torch.zeros(sum(self.numels), dtype=torch.float32, device=dev)

    def matches(self, tensors: Sequence[torch.Tensor]) -> bool:
        return [t.numel() for t in tensors] == self.numels and tensors[0].device == self.flat.device

    def all_reduce_mean_(self, tensors: Sequence[torch.Tensor]) -> None:
        ws = world_size()
        if not collectives_on():
            return
        views = self.flat.split(self.numels)
        torch._foreach_copy_(list(views), [t.reshape(-1) for t in tensors])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / ws)
        torch._foreach_copy_([t.reshape(-1) for t in tensors], list(views))


def all_reduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place all-reduce(sum) x 1/N of an already flat, contiguous buffer (``GaussianModel.grad_bucket``)."""
    ws = world_size()
    if collectives_on():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if ws > 1:
            flat.mul_(1.0 / ws)
    return flat


def all_reduce_max_(t: torch.Tensor) -> torch.Tensor:
    if collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier() -> None:
    if is_dist():
        dist.barrier()
