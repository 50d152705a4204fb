"""torch.optim.Adam as the reference builds it -- `torch.optim.Adam(model.get_param_groups(cfg.train), betas=(0.9, 0.999))`, train.py:263-267,
learning rates rewritten per iteration by update_lr (train.py:166-175) -- as ONE native launch per step (`gom_adam_multi`,
csrc/frame_parallel.hip) instead of torch's ~35 multi-tensor launches (0.58 ms of the drop-in Model's 4.4 ms iteration on MI355X).

Drop-in: the same constructor arguments (params or param groups with per-group `lr`, `name` and any other keys; `betas`, `eps`), the same
`param_groups` / `state` layout (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), so `state_dict()` / `load_state_dict()` exchange
checkpoints with torch.optim.Adam (formats.load_checkpoint), `zero_grad`, and `update_lr` keeps writing `param_group["lr"]`.
Not supported (the reference uses none of them): weight_decay, amsgrad, maximize, sparse or non-fp32 or host parameters -- those raise.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch


class GomAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False,
                 maximize: bool = False, capturable: bool = False):
        """capturable: the step count lives in device memory (as torch's capturable Adam), so `step()` can be captured into a HIP graph and
        replayed (train_util.GraphedTrainStep); learning rates are then those of the capture."""
        if weight_decay != 0.0 or amsgrad or maximize:
            raise NotImplementedError("GomAdam: weight_decay / amsgrad / maximize are not on GoMAvatar's path (train.py:263-267)")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0:
            raise ValueError("GomAdam: bad betas / eps")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=bool(capturable), differentiable=False, fused=None))
        from . import _lib
        self._lib = _lib
        self._step_dev: Optional[torch.Tensor] = None

    def _state_of(self, p: torch.Tensor) -> dict:
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)          # (torch keeps it on the host unless capturable)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = self._lib.load()
        # tensors that share betas / eps / step count go into one native call (the reference has ONE setting for all groups)
        calls = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse or not p.is_contiguous():
                    raise RuntimeError("GomAdam: contiguous fp32 device parameters with dense gradients only (there is no CPU path)")
                st = self._state_of(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (group["betas"], group["eps"], float(st["step"]), bool(group.get("capturable", False)))
                calls.setdefault(key, []).append((p, g, st, float(group["lr"])))
        if any(k[3] for k in calls) and len(calls) > 1:
            # ONE device counter serves a captured step: parameters with different step counts (a module that received its first gradient later) or
            # different betas / eps would each need their own -- and each native call would advance the shared one
            raise RuntimeError("GomAdam(capturable=True): all parameters must share betas, eps and step count (got %d different settings); "
                               "step the late parameters with a second optimizer, or use capturable=False" % len(calls))
        for (betas, eps, t, capt), items in calls.items():
            n = len(items)
            P = (ctypes.c_void_p * n)(*[i[0].data_ptr() for i in items])
            G = (ctypes.c_void_p * n)(*[i[1].data_ptr() for i in items])
            M = (ctypes.c_void_p * n)(*[i[2]["exp_avg"].data_ptr() for i in items])
            V = (ctypes.c_void_p * n)(*[i[2]["exp_avg_sq"].data_ptr() for i in items])
            N = (ctypes.c_int64 * n)(*[i[0].numel() for i in items])
            LR = (ctypes.c_float * n)(*[i[3] for i in items])
            sd = None
            if capt:
                if self._step_dev is None:
                    self._step_dev = torch.full((1,), int(t), dtype=torch.int64, device=items[0][0].device)
                sd = self._step_dev.data_ptr()
            self._lib.check(lib.gom_adam_multi(n, P, G, M, V, N, LR, int(t) + 1, sd, float(betas[0]), float(betas[1]), float(eps), self._lib.stream_ptr()))
            for i in items:          # (only a step that was enqueued counts; under graph REPLAY the device counter runs ahead of these: state_dict() reads it back)
                i[2]["step"] += 1
        return loss

    def _sync_steps(self) -> None:
        """capturable: the authoritative step count is the device counter (a graph replay advances it without running this Python); copy it into
        every state entry.  Synchronises the device: never call during capture."""
        if self._step_dev is not None:
            t = float(int(self._step_dev.item()))
            for st in self.state.values():
                if "step" in st:
                    st["step"] = torch.tensor(t, dtype=torch.float32)

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        """torch.optim.Adam checkpoints load unchanged; a capturable torch checkpoint holds `step` as DEVICE tensors -- moved to the host here (reading
        them inside step() would synchronise, which a capture forbids) -- and the device counter (kept: a captured step holds its address) is re-seeded from the loaded count."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get("step")) and st["step"].is_cuda:
                st["step"] = st["step"].detach().float().cpu()
        if self._step_dev is not None:
            # a captured step (GraphedTrainStep) has this tensor's address baked into its graph: keep the tensor, re-seed its value from the loaded count
            steps = {int(float(st["step"])) for st in self.state.values() if "step" in st}
            self._step_dev.fill_(max(steps) if steps else 0)
