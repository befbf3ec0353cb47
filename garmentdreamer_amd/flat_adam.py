"""``torch.optim.Adam`` for MANY SMALL tensors as one launch -- the optimizer of the NeTF stage's LoRA UNet.

The reference steps ``torch.optim.Adam(params, lr=unet_lr)`` over the 256 rank-4 adapter matrices, the camera
embedding and three shading embeddings (Garment_Deformer_NeTF/netf/trainer.py:129-137, stepped at :256).  On
MI355X that step is host work, not device work: the foreach implementation walks ~260 parameters in Python and
issues a dozen multi-tensor launches, and the GPU sits idle for ~3 ms of a 42 ms iteration while it does
(``tools/vsd_gaps.sh``: 2.4 ms between the last kernel of the backward pass and the first of the optimizer,
0.55 ms inside it; and the gap is not the optimizer's arithmetic but the host handling 260 gradient tensors after the
backward graph).  Here the fp32 parameters are re-seated as views of ONE flat buffer and their ``.grad`` are, for
good, views of ONE flat gradient buffer: the LoRA backward kernels add the adapter gradients straight into those
views (``Parameter._gd_grad_sink``, read by nn_ops._LoraLinear / _LoraBranch; ``gd_nn_lora_colreduce_pair_into``)
and return nothing to autograd; any other producer's gradient is accumulated in place by autograd as usual.  A step
is then ONE ``gd_scene_adam_step`` launch (include/gd_scene.h, the kernel the Gaussian scene's optimizer uses: same
update rule, same double-precision bias-correction scalars as torch's) and ``zero_grad`` ONE memset.

Semantics: ``.grad`` accumulates over backward passes until ``zero_grad()``, as in torch.  One difference, stated:
``torch.optim.Adam`` skips a parameter whose ``.grad`` is None (no moment decay, its own step count); a flat-set
parameter always has a gradient here (zeros if nothing wrote one), so its moments decay on every step.  That is only
the reference's behaviour for tensors that DO receive a gradient on every step, so the flat set is opt-in: the caller
names it (``flat=`` -- the rank-4 adapters, every one of which is on the path of every UNet call; ``for_lora_unet``
picks exactly those).  Everything else -- the camera MLP, the three shading embeddings of which one is used per step
(lora_unet.py:632-645), anything not fp32 or not on the GPU -- is stepped by a plain ``torch.optim.Adam`` over just
those few tensors, with torch's skipping and per-parameter step counts (round 5: for fp32 GPU tensors by the same HIP
kernel, one small launch per tensor that has a gradient -- same update rule, no foreach host work).

``param_groups`` is ONE list holding ONE dict, created once: ``for g in opt.param_groups: g["lr"] = x`` changes the
learning rate of both halves, as with a torch optimizer; ``step()`` reads lr / betas / eps from that dict.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import torch

from . import _native


class FlatAdam:
    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 flat: Optional[Iterable[torch.Tensor]] = None, exclude: Iterable[torch.Tensor] = (),
                 check_views: bool = True):
        """``flat``: the parameters (a subset of ``params``) that receive a gradient on EVERY step and may therefore live in
        the flat buffer -- those among them that are fp32 GPU tensors and not in ``exclude`` are re-seated NOW: construct the
        optimizer before any hipGraph that reads them is captured (a graph keeps the addresses it was captured with).
        ``flat=None``: nothing is re-seated, every parameter is stepped by ``torch.optim.Adam``."""
        self.params: List[torch.Tensor] = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        self._group = {"params": self.params, "lr": float(lr), "betas": (float(betas[0]), float(betas[1])), "eps": float(eps)}
        self._groups = [self._group]
        self.step_count = 0
        self._check_views = bool(check_views)    # a Python walk over the flat set per step (~50 us for 256 tensors)
        skip = {id(p) for p in exclude}
        listed = {id(p) for p in self.params}
        want = [] if flat is None else [p for p in flat]
        if any(id(p) not in listed for p in want):
            raise ValueError("FlatAdam: `flat` must be a subset of `params`")
        flat = [p for p in want if p.is_cuda and p.dtype == torch.float32 and id(p) not in skip]
        ids = {id(p) for p in flat}
        rest = [p for p in self.params if id(p) not in ids]
        self._flat_set = flat
        # the others: fp32 GPU tensors keep torch's semantics (skipped while .grad is None, their own step count) but are
        # stepped by the same HIP kernel, one small launch per tensor THAT HAS A GRADIENT (the NeTF stage: the camera MLP's
        # four tensors and the one shading embedding of the step) -- torch.optim.Adam's foreach path over them was ~0.4 ms of
        # host work with the GPU idle between the backward pass and the optimizer (profiles/r04_bench_vsd_gaps.txt);
        # anything else (bf16, CPU) goes to torch.optim.Adam
        self._solo = [p for p in rest if p.is_cuda and p.dtype == torch.float32]
        self._solo_state = {}
        solo_ids = {id(p) for p in self._solo}
        rest = [p for p in rest if id(p) not in solo_ids]
        self._rest = torch.optim.Adam(rest, lr=self.lr, betas=self.betas, eps=self.eps) if rest else None
        if not flat:
            return
        dev = flat[0].device
        if any(p.device != dev for p in flat):
            raise ValueError("FlatAdam: the fp32 parameters must live on one device")
        pad = lambda k: (k + 63) // 64 * 64      # every view starts 256-byte aligned (the LoRA kernels read 16-byte vectors)
        n = sum(pad(p.numel()) for p in flat)
        self._flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._grad = torch.zeros(n, dtype=torch.float32, device=dev)       # padding stays 0: its update is 0 / (0 + eps)
        self._exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._grad_views = []
        off = 0
        with torch.no_grad():
            for p in flat:
                k = p.numel()
                view = self._flat[off:off + k].view(p.shape)
                view.copy_(p.data)
                p.data = view                     # the module keeps its Parameter objects; their storage is the flat buffer
                gview = self._grad[off:off + k].view(p.shape)
                p.grad = gview                    # for good: zero_grad() clears the buffer, it does not drop the views
                p._gd_grad_sink = gview           # the LoRA backward kernels add into it (nn_ops._grad_sinks)
                self._grad_views.append(gview)
                off += pad(k)
        self._ends = (C.c_int64 * 1)(n)

    # ---- torch.optim.Optimizer surface the training loops use ---------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        if self._flat_set:
            self._grad.zero_()
        for p in self._solo:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()
        if self._rest is not None:
            self._rest.zero_grad(set_to_none=set_to_none)

    @classmethod
    def for_lora_unet(cls, unet, params: Iterable[torch.Tensor], **kw):
        """The optimizer of ``sd21.LoraUNet2DConditionModel``: its rank-4 adapters (``unet.lora_layers``, on the path of every
        UNet call) in the flat set, the camera MLP and the shading embeddings stepped with torch's skip semantics."""
        return cls(params, flat=list(unet.lora_layers.parameters()), **kw)

    @property
    def param_groups(self):
        return self._groups

    # lr / betas / eps live in the one group dict (what a scheduler or ``g["lr"] = x`` edits)
    @property
    def lr(self) -> float:
        return float(self._group["lr"])

    @lr.setter
    def lr(self, v: float):
        self._group["lr"] = float(v)

    @property
    def betas(self):
        b = self._group["betas"]
        return float(b[0]), float(b[1])

    @property
    def eps(self) -> float:
        return float(self._group["eps"])

    @torch.no_grad()
    def step(self):
        from . import nn_ops
        pending = nn_ops.lora_groups_pending()
        if pending:
            # grouped weight-gradient launch (nn_ops.LoraGradGroup) that never happened: a backward pass did not reach every
            # adapted projection of its forward pass -- stepping now would use gradients that miss those adapters
            raise RuntimeError(f"FlatAdam.step: {pending} LoRA weight-gradient problems were recorded but never launched")
        self.step_count += 1
        flat = self._flat_set
        if flat:
            if self._check_views:       # someone replaced a .grad (e.g. a foreign zero_grad(set_to_none=True)): put the view back
                for p, v in zip(flat, self._grad_views):
                    if p.grad is not v:
                        if p.grad is not None:
                            v.add_(p.grad)
                        p.grad = v
            lrs = (C.c_double * 1)(self.lr)
            dev = self._flat.device
            with torch.cuda.device(dev):
                _native.check_scene(_native.lib().gd_scene_adam_step(
                    torch.cuda.current_stream(dev).cuda_stream, self._flat.data_ptr(), self._grad.data_ptr(),
                    self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr(), self._flat.numel(), 1, self._ends, lrs,
                    self.betas[0], self.betas[1], self.eps, self.step_count), "gd_scene_adam_step")
        for p in self._solo:
            g = p.grad
            if g is None:
                continue                      # torch.optim.Adam skips it: no moment decay, its step count stands still
            st = self._solo_state.get(id(p))
            if st is None:
                st = self._solo_state[id(p)] = [0, torch.zeros_like(p, memory_format=torch.contiguous_format),
                                                torch.zeros_like(p, memory_format=torch.contiguous_format)]
            if not (p.is_contiguous() and g.is_contiguous() and g.dtype == torch.float32):
                raise RuntimeError("FlatAdam: fp32 parameters and their gradients must be contiguous fp32 tensors")
            st[0] += 1
            n = p.numel()
            with torch.cuda.device(p.device):
                _native.check_scene(_native.lib().gd_scene_adam_step(
                    torch.cuda.current_stream(p.device).cuda_stream, p.data_ptr(), g.data_ptr(), st[1].data_ptr(),
                    st[2].data_ptr(), n, 1, (C.c_int64 * 1)(n), (C.c_double * 1)(self.lr), self.betas[0], self.betas[1],
                    self.eps, st[0]), "gd_scene_adam_step")
        if self._rest is not None:
            for grp in self._rest.param_groups:
                grp["lr"], grp["betas"], grp["eps"] = self.lr, self.betas, self.eps
            self._rest.step()

    @property
    def flat_grad(self) -> Optional[torch.Tensor]:
        """The gathered gradient of the flat set as ONE tensor (what a data-parallel all-reduce would send)."""
        return getattr(self, "_grad", None)
