"""Optimiser step of DreamGaussian's Gaussians on the MI355X.

`FusedAdam` IS `torch.optim.Adam` (same constructor, same `state` / `param_groups`, so
`GaussianModel.training_setup` (gs_renderer.py:356-374) can build it with the same argument list and the
optimiser-state surgery of `densify_and_prune` (`replace_tensor_to_optimizer`, `_prune_optimizer`,
`cat_tensors_to_optimizer`, gs_renderer.py:464-545) keeps working) -- only `step()` differs: the ~8 elementwise
launches per parameter group of torch's default path become ONE launch over all groups (`gsr_adam_step`,
csrc/gsr_optim.hip), with torch's arithmetic in torch's order.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if weight_decay != 0 or amsgrad or kw.get("maximize") or kw.get("capturable") or kw.get("differentiable"):
            raise ValueError("FusedAdam covers what GaussianModel.training_setup uses: no weight decay, amsgrad, maximize, "
                             "capturable or differentiable")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        by_key = {}                                    # (device, step, beta1, beta2, eps) -> [(p, grad, state, lr)]
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda":
                    raise RuntimeError("FusedAdam runs on the GPU only (no CPU fallback); got a parameter on " + str(p.device))
                if p.grad.is_sparse or p.dtype is not torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs dense contiguous float32 parameters")
                st = self.state[p]
                if len(st) == 0:                       # torch's own lazy initialisation (adam.py: _init_group)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                key = (p.device, int(st["step"].item()), float(beta1), float(beta2), float(group["eps"]))
                by_key.setdefault(key, []).append((p, p.grad.contiguous(), st, float(group["lr"])))
        for (dev, step, beta1, beta2, eps), items in by_key.items():
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for i0 in range(0, len(items), 8):
                    chunk = items[i0:i0 + 8]
                    arr = (_lib.GsrAdamTensor * len(chunk))()
                    for j, (p, g, st, lr) in enumerate(chunk):
                        if not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
                        arr[j] = _lib.GsrAdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                    p.numel(), lr)
                    _lib.check(lib.gsr_adam_step(len(chunk), arr, step, beta1, beta2, eps, stream), "gsr_adam_step")
        return loss
