"""Frame-parallel data parallelism: one process per GPU, one frame per GPU per
step, ONE all-reduce of a flat fp32 gradient buffer (RCCL over xGMI on MI355X;
`gloo` in the CPU tests).

The reference has no distributed code at all; the path
shards naturally over frames because a frame's forward/backward only reads the
shared parameters.  The only exchange step is the gradient sum, so that is the
only collective.  The payload is small (3.8 MB at 55 104 Gaussians): latency-,
not bandwidth-bound, hence a single flat buffer instead of per-tensor calls.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class FlatBuffer:
    """Named tensors as views into one contiguous fp32 buffer."""

    def __init__(self, shapes: Sequence[Tuple[str, Tuple[int, ...]]], device, pad_to: int = 0):
        self.layout = []
        off = 0
        for name, shape in shapes:
            n = 1
            for d in shape:
                n *= int(d)
            self.layout.append((name, tuple(int(d) for d in shape), off, n))
            off += n
        self.numel = max(off, int(pad_to))
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.views: Dict[str, torch.Tensor] = {name: self.flat[o:o + n].view(shape) for name, shape, o, n in self.layout}

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.views[name]

    def items(self):
        return self.views.items()


class _DevicePtr:
    """A raw device allocation as a torch tensor (through __cuda_array_interface__): the peer-mapped gradient region of PeerAllReduce."""

    def __init__(self, ptr: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f4", "data": (int(ptr), False), "version": 3}


class PeerAllReduce:
    """The direct two-shot all-reduce of csrc/frame_parallel.hip (`gom_peer_reduce_*`): every rank's gradient buffer lives in a region
    the other ranks map through IPC handles (exchanged once over the process group); `run()` enqueues two kernels -- reduce-scatter in
    rank order, all-gather -- on the current stream.  One process per GPU; several processes on ONE device work too (the 1-GPU test)."""

    def __init__(self, numel: int, device, group: Optional[dist.ProcessGroup] = None, timeout_s: Optional[float] = None):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        self.lib = _lib.load()
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.numel = int(numel)
        torch.cuda.set_device(device)
        # Every step that can fail (allocation, IPC export, IPC mapping) is followed by an exchange of its outcome, so that ALL ranks
        # raise together instead of one leaving the others inside a collective.
        self._h, mine, err = None, None, None
        try:
            self._h = self.lib.gom_peer_reduce_create(self.rank, self.world, self.numel)
            if not self._h:
                _lib.check(-1)
            if timeout_s is not None:
                _lib.check(self.lib.gom_peer_reduce_set_timeout(self._h, float(timeout_s)))
            buf = (ctypes.c_ubyte * 64)()
            _lib.check(self.lib.gom_peer_reduce_handle(self._h, buf))
            mine = bytes(buf)
        except Exception as e:
            err = f"rank {self.rank}: {type(e).__name__}: {e}"
        got = [None] * self.world
        dist.all_gather_object(got, (err, mine), group=group)
        self._raise_together([g[0] for g in got])
        try:
            blob = (ctypes.c_ubyte * (64 * self.world)).from_buffer_copy(b"".join(g[1] for g in got))
            _lib.check(self.lib.gom_peer_reduce_connect(self._h, blob))
            self.buffer = torch.as_tensor(_DevicePtr(self.lib.gom_peer_reduce_buffer(self._h), self.numel), device=torch.device(device))
        except Exception as e:
            err = f"rank {self.rank}: {type(e).__name__}: {e}"
        got = [None] * self.world
        dist.all_gather_object(got, err, group=group)   # (also the barrier: every rank has mapped every region before anyone raises a flag in it)
        self._raise_together(got)

    def _raise_together(self, errs) -> None:
        bad = [e for e in errs if e]
        if bad:
            self.close(collective=False)   # (nobody has raised a flag in anybody's region yet)
            raise RuntimeError("peer all-reduce unavailable (" + bad[0] + ")")

    def poll(self) -> None:
        """Every step, before the exchange is enqueued: raises if a wait of an EARLIER exchange gave up (the kernel that times out writes a
        pinned host word; reading it costs no synchronisation).  The native run calls refuse to enqueue on a failed handle as well."""
        if self.lib.gom_peer_reduce_poll(self._h):
            raise RuntimeError("peer all-reduce: a peer did not answer within the wait limit -- the gradients of that step were NOT reduced "
                               "(reset() on every rank, or rebuild the exchange, before stepping again)")

    def run(self, out: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
        out = self.buffer if out is None else out
        assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == self.numel
        self.poll()
        self._lib.check(self.lib.gom_peer_reduce_run(self._h, out.data_ptr(), float(scale), self._lib.stream_ptr()))
        return out

    def run_adam(self, opt: "FlatAdam", scale: float = 1.0, out: Optional[torch.Tensor] = None, zero1: bool = False) -> None:
        """The exchange with `opt`'s Adam step inside it.  Default (`gom_peer_reduce_run_adam`): every rank updates its parameter replica
        straight from the reduced slices in the all-gather; `out` (optional) receives the reduced gradient.  zero1 (`gom_peer_reduce_run_zero1`,
        SURVEY.md 8(e)): the rank that reduced a slice steps THAT slice and the all-gather moves parameters -- the same bits, 1 / world of
        the optimizer arithmetic and of the moment traffic per rank."""
        self.poll()
        lr = (self._ct.c_float * len(opt.lr))(*opt.lr)
        P = self._lib.ptr
        if zero1:
            assert out is None, "the ZeRO-1 exchange gathers parameters: there is no reduced gradient to hand out"
            self._lib.check(self.lib.gom_peer_reduce_run_zero1(self._h, float(scale), P(opt.fp.params.flat), P(opt.exp_avg), P(opt.exp_avg_sq), len(opt.lr),
                                                               opt._begin, lr, opt.t + 1, opt.betas[0], opt.betas[1], opt.eps, self._lib.stream_ptr()))
        else:
            self._lib.check(self.lib.gom_peer_reduce_run_adam(self._h, float(scale), P(out), P(opt.fp.params.flat), P(opt.exp_avg), P(opt.exp_avg_sq), len(opt.lr),
                                                              opt._begin, lr, opt.t + 1, opt.betas[0], opt.betas[1], opt.eps, self._lib.stream_ptr()))
        opt.t += 1   # (only a step that was enqueued counts)

    def check(self) -> None:
        """Synchronises; raises if a peer never answered (the kernels give up after the wait limit instead of hanging)."""
        self._lib.check(-self.lib.gom_peer_reduce_status(self._h))

    def reset(self) -> None:
        """After a timeout, on EVERY rank: clears the condition between two barriers of the group and moves all ranks to a common epoch."""
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        ep = torch.tensor([int(self.lib.gom_peer_reduce_epoch(self._h))], dtype=torch.int64)
        got = [None] * self.world
        dist.all_gather_object(got, int(ep.item()), group=self.group)
        self._lib.check(self.lib.gom_peer_reduce_reset(self._h, max(got) & 0xffffffff))
        dist.barrier(group=self.group)

    def close(self, collective: bool = True) -> None:
        """Frees the region.  A peer's all-gather may still be reading this rank's reduced slice: the device is synchronised and the group
        passes a barrier first (collective=False only where no exchange can be in flight)."""
        if getattr(self, "_h", None):
            self.buffer = None
            if collective and dist.is_initialized():
                try:
                    torch.cuda.synchronize()
                    dist.barrier(group=self.group)
                except Exception:   # (interpreter shutdown, a peer already gone: free anyway)
                    pass
            self.lib.gom_peer_reduce_destroy(self._h)
            self._h = None


class FrameParallel:
    """Replicated parameters + flat gradient buffer + one all-reduce per step.

    `shapes` lists (name, shape) of every trainable tensor.  Gradients are
    written (by the HIP pipeline or by autograd) into `self.grads[name]`, which
    are views of `self.grads.flat`; `all_reduce_grads()` sums them over ranks
    (and divides by the world size when `average`), after which every rank
    applies the same optimizer step and stays bit-identical.
    """

    def __init__(self, shapes: Sequence[Tuple[str, Tuple[int, ...]]], device, group: Optional[dist.ProcessGroup] = None,
                 average: bool = True, pad_to: int = 0, impl: str = "collective"):
        """impl: "collective" = torch.distributed all_reduce (RCCL on GPUs, gloo on the host); "peer" = the direct two-shot all-reduce
        over IPC-mapped peer buffers (`PeerAllReduce`; the gradient buffer then lives in the peer-mapped region); "peer-zero1" = the same
        exchange with the optimizer sharded over the ranks (`all_reduce_and_step` only: each rank steps its slice, parameters are gathered)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.average = average
        self._nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self._native_avg = self._nccl
        self._host = None
        self.params = FlatBuffer(shapes, device)
        self.grads = FlatBuffer(shapes, device, pad_to=pad_to)
        self.impl, self.peer = impl, None
        if impl not in ("collective", "peer", "peer-zero1"):
            raise ValueError(f"unknown all-reduce implementation {impl!r}")
        if impl in ("peer", "peer-zero1") and self.world > 1:
            self.peer = PeerAllReduce(self.grads.numel, device, group)
            self.grads.flat = self.peer.buffer          # gradients are written straight into the peer-mapped region
            self.grads.views = {name: self.grads.flat[o:o + n].view(shape) for name, shape, o, n in self.grads.layout}

    def frame_index(self, step: int) -> int:
        """Global index of the frame this rank renders at `step`."""
        return step * self.world + self.rank

    def broadcast_params(self, src: int = 0) -> None:
        if self.world > 1:
            dist.broadcast(self.params.flat, src=src, group=self.group)

    def all_reduce_grads(self) -> None:
        """ONE collective on the flat buffer (enqueued on the current stream).  RCCL averages inside the collective
        (ReduceOp.AVG: no separate scaling launch); gloo (CPU tests, ranks sharing a device) sums and scales."""
        if self.world <= 1:
            return
        flat = self.grads.flat
        if self.peer is not None:
            self.peer.run(flat, 1.0 / self.world if self.average else 1.0)
            return
        if self._nccl:
            if self.average and self._native_avg:
                try:
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
                    return
                except (RuntimeError, ValueError):      # a collective library without ncclAvg: sum + scale from now on (same on every rank)
                    self._native_avg = False
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                flat.mul_(1.0 / self.world)
        elif flat.is_cuda:
            # gloo with device tensors (ranks sharing one GPU on a development lease): staged through one pinned host buffer --
            # gloo's plain host path -- rather than through its device-tensor path
            if self._host is None:
                self._host = torch.empty(flat.shape, dtype=torch.float32).pin_memory()
            self._host.copy_(flat, non_blocking=False)
            dist.all_reduce(self._host, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                self._host.mul_(1.0 / self.world)
            flat.copy_(self._host, non_blocking=False)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                flat.mul_(1.0 / self.world)

    def all_reduce_and_step(self, opt: "FlatAdam") -> None:
        """Mean of the gradients over the ranks + `opt`'s Adam step.  Over the peer exchange that is two launches (the optimizer rides in
        the all-gather); otherwise the collective followed by `opt.step()`."""
        if self.peer is not None and not opt.graphable:
            self.peer.run_adam(opt, 1.0 / self.world if self.average else 1.0, zero1=self.impl == "peer-zero1")
        else:
            self.all_reduce_grads()
            opt.step()

    def barrier_after_pause(self) -> None:
        """Call after a long host-side pause on SOME ranks only (checkpoint write, evaluation, subdivision + optimizer rebuild, first-step
        graph capture) and before the next exchange: the peer exchange's waits are bounded (30 s by default), a barrier makes the ranks meet
        on the host first, where waiting is free."""
        if self.world > 1:
            dist.barrier(group=self.group)

    def close(self) -> None:
        """Collective: releases the peer-mapped region behind a device synchronisation and a barrier of the group."""
        if self.peer is not None:
            self.grads.views, self.grads.flat = {}, None
            self.peer.close()
            self.peer = None

    def __del__(self):
        try:
            if getattr(self, "peer", None) is not None:
                self.peer.close(collective=False)   # (garbage collection is not a collective point: close() is the orderly way)
        except Exception:
            pass

    def make_adam(self, lrs: Dict[str, float], betas=(0.9, 0.999), eps: float = 1e-8) -> torch.optim.Adam:
        """Adam with one param group per tensor (the reference uses per-group
        learning rates, models/model.py:305-324), state living next to the flat buffers."""
        groups = []
        for name, p in self.params.items():
            p.requires_grad_(True)
            p.grad = self.grads[name]
            groups.append({"params": [p], "lr": float(lrs.get(name, lrs.get("default", 1e-3))), "name": name})
        return torch.optim.Adam(groups, betas=betas, eps=eps)


class FlatAdam:
    """torch.optim.Adam(param_groups, betas, eps) -- the reference's optimizer, train.py:263-267 -- as ONE native launch over the flat
    parameter buffer (`gom_adam_flat`, csrc/frame_parallel.hip): per-tensor learning rates (`lrs[name]`, or `lrs['default']`), moments
    in two more flat buffers, the step count on the host.  Device buffers only: there is no CPU path (the gloo tests use `make_adam`)."""

    def __init__(self, fp: "FrameParallel", lrs: Dict[str, float], betas=(0.9, 0.999), eps: float = 1e-8, graphable: bool = False,
                 lr_decay_steps: float = 0.0):
        """graphable: the step count lives in device memory and the learning-rate decay (update_lr) is derived from it in the kernel
        (`gom_adam_flat_graphable`), so `step()` can be captured into a graph behind the frame step and replayed."""
        import ctypes
        from . import _lib
        if not fp.params.flat.is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU only (libgom_hip.so); use FrameParallel.make_adam on the host")
        self.fp, self._lib, self._ct = fp, _lib, ctypes
        self.betas, self.eps, self.t = (float(betas[0]), float(betas[1])), float(eps), 0
        n = fp.params.numel
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=fp.params.flat.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.names = [name for name, _, _, _ in fp.params.layout]
        bounds = [off for _, _, off, _ in fp.params.layout] + [fp.params.layout[-1][2] + fp.params.layout[-1][3]]
        self._begin = (ctypes.c_int64 * len(bounds))(*bounds)
        self.base_lr = [float(lrs.get(name, lrs.get("default", 1e-3))) for name in self.names]
        self.lr = list(self.base_lr)
        self.graphable, self.lr_decay_steps = bool(graphable), float(lr_decay_steps)
        self.step_dev = torch.zeros(2, dtype=torch.int64, device=fp.params.flat.device) if graphable else None

    def attach(self, state, grad_scale: float = 1.0) -> None:
        """Make this optimizer's step the LAST launch of `state`'s frame step (`gom_state_set_frame_optimizer`): forward, backward and Adam
        are then one recorded graph.  Needs `graphable=True` (the step count lives on the device); do not call `step()` as well."""
        assert self.graphable, "attach() needs FlatAdam(graphable=True)"
        if not self._lib.has_lab():
            raise RuntimeError("FlatAdam.attach needs a -DGOM_LAB build of libgom_hip.so (include/gom_hip_lab.h): measured slower than a plain launch "
                               "behind the frame step's graph, it is not part of the product library")
        fp, P = self.fp, self._lib.ptr
        lr = (self._ct.c_float * len(self.base_lr))(*self.base_lr)
        self._lib.check(self._lib.load().gom_state_set_frame_optimizer(state.handle, fp.params.numel, P(fp.params.flat), P(fp.grads.flat), P(self.exp_avg), P(self.exp_avg_sq),
                                                                       len(self.base_lr), self._begin, lr, P(self.step_dev), self.lr_decay_steps, self.betas[0], self.betas[1],
                                                                       self.eps, float(grad_scale)))

    def decay(self, iter_step: int, lr_decay_steps: float) -> None:
        """update_lr (train.py:166-175)."""
        self.lr = [b * 0.1 ** (iter_step / lr_decay_steps) for b in self.base_lr]

    def step(self, grad_scale: float = 1.0) -> None:
        """One Adam step on the current stream, reading `fp.grads.flat` (as the all-reduce left it)."""
        fp, P = self.fp, self._lib.ptr
        if self.graphable:
            lr = (self._ct.c_float * len(self.base_lr))(*self.base_lr)
            self._lib.check(self._lib.load().gom_adam_flat_graphable(fp.params.numel, P(fp.params.flat), P(fp.grads.flat), P(self.exp_avg), P(self.exp_avg_sq),
                                                                     len(self.base_lr), self._begin, lr, 1, P(self.step_dev), self.lr_decay_steps, self.betas[0],
                                                                     self.betas[1], self.eps, float(grad_scale), self._lib.stream_ptr()))
            self.t += 1
            return
        lr = (self._ct.c_float * len(self.lr))(*self.lr)
        self._lib.check(self._lib.load().gom_adam_flat(fp.params.numel, P(fp.params.flat), P(fp.grads.flat), P(self.exp_avg), P(self.exp_avg_sq), len(self.lr),
                                                       self._begin, lr, self.t + 1, self.betas[0], self.betas[1], self.eps, float(grad_scale), self._lib.stream_ptr()))
        self.t += 1   # (only a step that was enqueued counts)


def shapes_for_model(n_verts: int, n_faces: int, extra: Iterable[Tuple[str, Tuple[int, ...]]] = ()):
    """The hot path's trainables in the reference's layouts (models/model.py:74-85,
    appearance_module.py:14) followed by any extra tensors (MLP weights...)."""
    return [("vertices", (3, n_verts)), ("so3", (3, n_faces)), ("scale", (3, n_faces)), ("appearance", (3, n_faces))] + list(extra)
