"""Adam over all parameter groups in one launch.

`FusedAdam(param_groups, lr=..., betas=..., eps=...)` takes what `initialize_optimizer` hands to `torch.optim.Adam`
(src/vtgaussian_slam.py:180-187: one group per tensor, `{'params': [v], 'name': k, 'lr': lrs[k]}`; defaults for
tracking, `lr=0.0, eps=1e-15` for mapping) and exposes the part of the optimizer interface the driver uses:
`step()`, `zero_grad(set_to_none=True)`, `param_groups`, `state`.  Same update rule as torch (no weight decay, no
amsgrad); parameters without a gradient are skipped, like torch does.  There is no CPU path.

`skip_frozen=True` additionally leaves the parameters of groups with lr == 0 alone: torch still streams them (the update
is p - 0 * step, and the moment estimates move), which at N = 1 M costs ~28 us per mapping iteration for the two frozen
groups of every shipped configuration (means3D and unnorm_rotations, 7 of 12 floats per Gaussian).  The parameters come out
bit-identical; only `state` of the frozen tensors stays empty, which matters to a caller that raises such a group's lr later
on the same optimizer object (the reference builds a new optimizer per phase, src/vtgaussian_slam.py:180-187)."""
from __future__ import annotations

import ctypes
from typing import Dict, Iterable, List

import torch

from . import _I32, _P, _check, _lib, _stream_ptr

_MAX = 8


class _Group(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("count", ctypes.c_uint64), ("lr", ctypes.c_float), ("eps", ctypes.c_float)]


_lib.vtgs_adam_step.restype = ctypes.c_int
_lib.vtgs_adam_step.argtypes = [ctypes.POINTER(_Group), _I32, _I32, ctypes.c_float, ctypes.c_float, _P]
_lib.vtgs_adam_step_rows.restype = ctypes.c_int
_lib.vtgs_adam_step_rows.argtypes = [ctypes.POINTER(_Group), _I32, _I32, ctypes.c_float, ctypes.c_float, _P, _I32, _I32, _P]


class FusedAdam:
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, skip_frozen: bool = False):
        self.skip_frozen = bool(skip_frozen)
        params = list(params)
        if params and not isinstance(params[0], dict):
            params = [{"params": params}]
        self.defaults = {"lr": lr, "betas": tuple(betas), "eps": eps}
        self.param_groups: List[Dict] = []
        for g in params:
            g = dict(g)
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            self.param_groups.append(g)
        self.state: Dict[torch.Tensor, Dict] = {}

    def zero_grad(self, set_to_none: bool = True) -> None:
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_().zero_()

    @torch.no_grad()
    def step(self, rows: torch.Tensor = None) -> None:
        """`rows` (int32 device tensor of row indices, `partition.OwnerExchange.update_rows`): the per-Gaussian tensors -- those
        whose first dimension equals that of the first parameter group -- are updated on those rows only (parameter and both
        moments; vtgs_adam_step_rows); every other tensor (the camera poses) as usual."""
        n_total = None
        if rows is not None:
            if rows.dtype != torch.int32 or not rows.is_cuda or not rows.is_contiguous():
                raise TypeError("FusedAdam.step(rows=): an int32 contiguous tensor on the HIP device")
            n_total = int(self.param_groups[0]["params"][0].shape[0])
        # (step, betas) buckets: parameters that first received a gradient at different iterations have different
        # bias corrections, exactly as with torch's per-parameter step counters.
        buckets: Dict[tuple, List[_Group]] = {}
        keep = []                                                    # tensors the launch reads must outlive the enqueue
        for g in self.param_groups:
            if self.skip_frozen and float(g["lr"]) == 0.0:
                continue
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam needs parameters on a HIP device (torch 'cuda'); no CPU path exists")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("FusedAdam: float32 contiguous parameters only")
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                st["step"] += 1
                grad = p.grad.to(torch.float32).contiguous()
                keep.append(grad)
                rec = _Group(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             p.numel(), float(g["lr"]), float(g["eps"]))
                by_rows = n_total is not None and p.dim() >= 1 and int(p.shape[0]) == n_total
                buckets.setdefault((st["step"], tuple(g["betas"]), p.device, by_rows), []).append(rec)
        for (step, betas, dev, by_rows), recs in buckets.items():
            for i in range(0, len(recs), _MAX):
                chunk = recs[i:i + _MAX]
                arr = (_Group * len(chunk))(*chunk)
                if by_rows:
                    _check(_lib.vtgs_adam_step_rows(arr, len(chunk), int(step), float(betas[0]), float(betas[1]), rows.data_ptr(),
                                                    int(rows.numel()), n_total, _stream_ptr(dev)), "vtgs_adam_step_rows")
                else:
                    _check(_lib.vtgs_adam_step(arr, len(chunk), int(step), float(betas[0]), float(betas[1]), _stream_ptr(dev)),
                           "vtgs_adam_step")
