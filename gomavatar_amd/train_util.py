"""`unpack` and `compute_loss` of the reference's training loop (train.py:53-55, 98-163) on top of the HIP path, plus
the three mesh regularisers it calls (PyTorch3D `mesh_laplacian_smoothing(method="uniform")`,
`mesh_normal_consistency`, utils/network_util.py `mesh_color_consistency`) written against `model.SimpleMesh`."""
from __future__ import annotations

import os
import torch
import torch.nn.functional as F

from .model import _get


def unpack(rgbs, masks, bgcolors):
    """train.py:53-55."""
    if rgbs.is_cuda and rgbs.dtype == torch.float32 and rgbs.dim() == 4 and rgbs.shape[-1] == 3 and bgcolors.dim() == 2 and not bgcolors.requires_grad:
        from .losses import unpack_fused
        return unpack_fused(rgbs, masks, bgcolors)            # one kernel each way (csrc/loss.hip)
    return rgbs * masks.unsqueeze(-1) + bgcolors[:, None, None, :] * (1 - masks).unsqueeze(-1)


def mesh_laplacian_smoothing(mesh, reduce: bool = True) -> torch.Tensor:
    """utils/network_util.py:669-792 (uniform): mean over vertices of || (1/deg) sum_neighbours v_j - v_i ||^2
    (`loss.norm(dim=1) ** 2` then the un-weighted mean, :789-792); no gradient through the Laplacian matrix."""
    v, e = mesh.verts_packed(), mesh.edges_packed()
    if v.is_cuda and getattr(mesh, "loss_topo", None) is not None:
        from .mesh_losses import laplacian_smoothing
        return laplacian_smoothing(v, mesh.loss_topo, reduce)
    N = v.shape[0]
    deg = torch.zeros(N, device=v.device, dtype=v.dtype).index_add(0, e[:, 0], torch.ones(e.shape[0], device=v.device, dtype=v.dtype))
    deg = deg.index_add(0, e[:, 1], torch.ones(e.shape[0], device=v.device, dtype=v.dtype))
    s = torch.zeros_like(v).index_add(0, e[:, 0], v[e[:, 1]]).index_add(0, e[:, 1], v[e[:, 0]])
    lap = s / deg.clamp_min(1.0)[:, None] - v
    return (lap.norm(dim=1) ** 2).mean()


def mesh_normal_consistency(mesh, face_connectivity=None, reduce: bool = True) -> torch.Tensor:
    """PyTorch3D mesh_normal_consistency(mesh) as called at train.py:149: 1 - cos between the normals of EVERY pair of faces
    sharing an edge (cosine_similarity eps 1e-8), averaged over the pairs.  `face_connectivity` (optional, host path only)
    overrides the pair list; the default is every edge-adjacent pair of the mesh (`mesh.normal_pairs`)."""
    v, f = mesh.verts_packed(), mesh.faces_packed()
    if v.is_cuda and getattr(mesh, "loss_topo", None) is not None and face_connectivity is None:
        from .mesh_losses import normal_consistency
        return normal_consistency(v, mesh.topo, mesh.loss_topo, reduce)
    pairs = face_connectivity if face_connectivity is not None else mesh.normal_pairs
    n = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1)
    a, b = n[pairs[:, 0]], n[pairs[:, 1]]
    w = (a * a).sum(1) * (b * b).sum(1)
    return (1.0 - (a * b).sum(1) / torch.sqrt(w.clamp_min(1e-16))).mean()


def mesh_color_consistency(colors, face_connectivity, loss_topo=None, reduce: bool = True) -> torch.Tensor:
    """network_util.py:795-799: mean absolute colour difference of edge-adjacent faces."""
    if colors.is_cuda and loss_topo is not None:
        from .mesh_losses import color_consistency
        return color_consistency(colors.T, loss_topo, reduce)      # (F,3) view of the (3,F) parameter -> its own layout
    return (colors[face_connectivity[:, 0]] - colors[face_connectivity[:, 1]]).abs().mean()


_COEFF_CACHE = {}
_REF_ORDER = ("rgb", "mask", "lpips", "laplacian_canoincal", "laplacian_observation", "normal_mask", "normal_consist", "color_consist")


def compute_loss(rgb_pred, mask_pred, outputs, rgb_gt, mask_gt, loss_cfg, data=None, i_iter=0, tb=None, lpips_func=None, **kwargs):
    """train.py:98-163.  `lpips_func(pred_nchw_pm1, gt_nchw_pm1)` as in the reference, or an object with `.loss(pred_nhwc01,
    gt_nhwc01)` (gomavatar_amd.lpips.LPIPSMatrixCore)."""
    names, values, coeffs = [], [], []

    def put(name, value, coeff):
        names.append(name); values.append(value); coeffs.append(float(coeff))

    want_nm = _get(loss_cfg, "normal.coeff_mask", 0.0) > 0 and outputs.get("normal_mask") is not None
    k = int(_get(loss_cfg, "normal.kernel_size", 5))
    fused = rgb_pred.is_cuda and rgb_pred.shape[0] == 1 and rgb_pred.dtype == torch.float32 and (k % 2 == 1 or not want_nm)
    if fused:      # the three mean-|a - b| terms in one kernel each way (csrc/loss.hip), the dilation of the target mask included
        from .losses import l1_terms
        l1 = l1_terms(rgb_pred, rgb_gt, mask_pred, mask_gt, outputs["normal_mask"] if want_nm else None, k if want_nm else 0)
        put("rgb", None, _get(loss_cfg, "rgb.coeff", 1.0))
        put("mask", None, _get(loss_cfg, "mask.coeff", 5.0))
    else:
        put("rgb", torch.mean(torch.abs(rgb_pred - rgb_gt)), _get(loss_cfg, "rgb.coeff", 1.0))
        put("mask", torch.mean(torch.abs(mask_pred - mask_gt)), _get(loss_cfg, "mask.coeff", 5.0))
    if want_nm:    # (the reference's dict order is rgb, mask, lpips, ..., normal_mask: only the summation order of `total` differs)
        if fused:
            put("normal_mask", None, loss_cfg.normal.coeff_mask)
        else:
            dil = F.max_pool2d(mask_gt.unsqueeze(1), kernel_size=k, stride=1, padding=k // 2).squeeze(1)
            put("normal_mask", torch.mean(torch.abs(outputs["normal_mask"] - dil)), loss_cfg.normal.coeff_mask)
    # fused: every term hands over its PARTIAL sums ((1, n) matrices; reduce=False) and ONE launch folds them all (losses.loss_tail) instead of a
    # reduction per term, a concatenation, a multiply and a final sum (nine launches of ~4.5 us + ~2.5 us between launches, and their backward)
    tail = fused and os.environ.get("GOM_LOSS_TAIL", "1") != "0"   # (development switch: 0 = a reduction per term + cat + multiply + sum)
    mats = {}        # name -> (matrix of partial sums, factor in front of its sum)

    def term(name, fn, coeff, pre=1.0):
        v = fn(not tail)
        if tail and torch.is_tensor(v) and v.dim() == 2:
            mats[name] = (v, pre)
            v = None
        put(name, v, coeff)
    if lpips_func is not None and _get(loss_cfg, "lpips.coeff", 1.0) > 0:
        if hasattr(lpips_func, "loss"):
            def _lpips_term(r):
                if getattr(lpips_func, "supports_partial_sums", False):        # LPIPSMatrixCore: hands its partial sums to the fused tail
                    return lpips_func.loss(rgb_pred, rgb_gt, reduce=r)
                return lpips_func.loss(rgb_pred, rgb_gt)                        # a caller's own object with the plain loss(pred, gt) signature: the reduced scalar
            term("lpips", _lpips_term, _get(loss_cfg, "lpips.coeff", 1.0), 1.0 / rgb_pred.shape[0])
        else:
            put("lpips", torch.mean(lpips_func(2 * rgb_pred.permute(0, 3, 1, 2) - 1, 2 * rgb_gt.permute(0, 3, 1, 2) - 1)), _get(loss_cfg, "lpips.coeff", 1.0))
    if _get(loss_cfg, "laplacian.coeff_canonical", 0.0) > 0:
        term("laplacian_canoincal", lambda r: mesh_laplacian_smoothing(outputs["mesh_canonical"], reduce=r), loss_cfg.laplacian.coeff_canonical)
    if _get(loss_cfg, "laplacian.coeff_observation", 0.0) > 0:
        term("laplacian_observation", lambda r: mesh_laplacian_smoothing(outputs["mesh"], reduce=r), loss_cfg.laplacian.coeff_observation)
    if _get(loss_cfg, "normal.coeff_consist", 0.0) > 0:
        term("normal_consist", lambda r: mesh_normal_consistency(outputs["mesh"], reduce=r), loss_cfg.normal.coeff_consist)      # train.py:149: all edge-adjacent pairs
    if _get(loss_cfg, "color_consist.coeff", 0.0) > 0:
        term("color_consist", lambda r: mesh_color_consistency(outputs["colors"], outputs["face_connectivity"], getattr(outputs.get("mesh"), "loss_topo", None), reduce=r),
             loss_cfg.color_consist.coeff)
    if not fused:
        order = sorted(range(len(names)), key=lambda i: _REF_ORDER.index(names[i]))
        losses = {names[i]: {"unscaled": values[i], "scaled": values[i] * coeffs[i]} for i in order}
        return sum(item["scaled"] for item in losses.values()), losses
    key = (tuple(coeffs), rgb_pred.device)
    cvec = _COEFF_CACHE.get(key)
    if cvec is None:
        if len(_COEFF_CACHE) > 64:
            _COEFF_CACHE.clear()
        cvec = _COEFF_CACHE[key] = torch.tensor(coeffs, dtype=torch.float32, device=rgb_pred.device)
    order = sorted(range(len(names)), key=lambda i: _REF_ORDER.index(names[i]))      # the reference's dict order
    n_l1 = 3 if want_nm else 2                                                        # (the L1 terms are the first entries of `names`)
    if all(v is None for v in values):
        from .losses import loss_tail
        rest = [mats[n] for n in names[n_l1:]]
        vec, scaled, total = loss_tail(cvec, [l1.unsqueeze(1)] + [m for m, _ in rest], [n_l1] + [1] * len(rest), [1.0] + [p for _, p in rest])
        losses = {names[i]: {"unscaled": vec[i], "scaled": scaled[i]} for i in order}
        return total, losses
    # a term came as a scalar (a host path, a callable LPIPS): one vector of all terms, total = (vector * coefficients).sum() -- three launches
    # instead of a multiply and an add per term (and as many again backward); the dict entries are views into it, still differentiable
    rest = []
    for n, v in zip(names[n_l1:], values[n_l1:]):
        if v is None:
            m, p = mats[n]
            v = m.sum() if p == 1.0 else m.sum() * p
        rest.append(v.reshape(1))
    vec = torch.cat([l1[:n_l1]] + rest) if rest else l1[:n_l1]
    scaled = vec * cvec
    losses = {names[i]: {"unscaled": vec[i], "scaled": scaled[i]} for i in order}
    return scaled.sum(), losses


def update_lr(optimizer, iter_step, train_cfg) -> None:
    """train.py:166-175: lr = base * 0.1 ** (iter / lr_decay_steps), base = cfg.train.lr.<group name>."""
    decay_value = 0.1 ** (iter_step / train_cfg.lr_decay_steps)
    for param_group in optimizer.param_groups:
        if not hasattr(train_cfg.lr, param_group["name"]):
            raise AttributeError(f"no learning rate for parameter group {param_group['name']!r} (the reference's fallback reads an undefined cfg.train.train.lr)")
        param_group["lr"] = getattr(train_cfg.lr, param_group["name"]) * decay_value


_ONES = {}


def backward_from(loss: torch.Tensor) -> None:
    """loss.backward() with the seed gradient (a 1 of the loss's shape) kept per device instead of filled per iteration (one launch less)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones_like(loss)
    loss.backward(gradient=one)


def forward_backward(model, data, train_cfg, n_iters, lpips_func=None, random_bgcolor: bool = True):
    """train.py:317-338: forward -> unpack -> compute_loss -> backward (gradients accumulate into `.grad`).  Returns (loss, loss_items, rgb, mask)."""
    rgb, mask, outputs = model(data["K"], data["E"], data["cnl_gtfms"], data["dst_Rs"], data["dst_Ts"], dst_posevec=data.get("dst_posevec"),
                               canonical_joints=data.get("dst_tpose_joints"), i_iter=n_iters, bgcolor=data.get("bgcolor"))
    if random_bgcolor:
        rgb = unpack(rgb, mask, data["bgcolor"])
    loss, loss_items = compute_loss(rgb, mask, outputs, data["target_rgbs"], data["target_masks"], train_cfg.losses, data, n_iters, lpips_func=lpips_func)
    backward_from(loss)
    return loss, loss_items, rgb, mask


def train_iteration(model, optimizer, data, train_cfg, n_iters, lpips_func=None, random_bgcolor: bool = True, frame_parallel=None):
    """One iteration of the reference's loop, train.py:313-348 (without logging / checkpoints / subdivision, which the caller owns):
    zero_grad -> forward -> unpack -> compute_loss -> backward -> optimizer step -> update_lr.  Returns (loss, loss_items, rgb, mask).

    frame_parallel (parallel.ModelFrameParallel, `optimizer` may be None): the same iteration as ONE of N frame-parallel ranks
    (BASELINE configs[3]; `data` = this rank's frame, `frame_parallel.frame_index(step)`): gradients accumulate straight into the
    flat exchange buffer, ONE exchange forms their mean over the ranks with the reference's Adam + update_lr inside / behind it, on
    its own stream -- the next iteration's LPIPS target trunk is enqueued before the main stream waits for the exchange and runs under it."""
    if hasattr(lpips_func, "prefetch_target") and _get(train_cfg.losses, "lpips.coeff", 1.0) > 0:
        lpips_func.prefetch_target(data["target_rgbs"])       # (the target's half of the LPIPS trunk, on a second stream under the frame's forward)
    if frame_parallel is not None:
        frame_parallel.zero_grad()                            # (waits for the previous step's exchange: the parameters it leaves are this forward's)
    else:
        optimizer.zero_grad()
    out = forward_backward(model, data, train_cfg, n_iters, lpips_func, random_bgcolor)
    if frame_parallel is not None:
        frame_parallel.step(n_iters)
    else:
        optimizer.step()
        update_lr(optimizer, n_iters, train_cfg)
    return out


def eval_frame(model, data, bgcolor255=(255.0, 255.0, 255.0)):
    """eval.py:339-357: no_grad forward, composition on cfg.bgcolor / 255, 8-bit quantisation -> (H,W,3) uint8 (+ the PSNR of
    eval.py:101-104 against data['target_rgbs'] when present)."""
    from .metrics import from_8b, psnr, to_8b
    with torch.no_grad():
        pred, mask, _ = model(data["K"], data["E"], data["cnl_gtfms"], data["dst_Rs"], data["dst_Ts"], data.get("dst_posevec"))
        bg = torch.tensor(bgcolor255, dtype=torch.float32, device=pred.device)[None] / 255.0
        pred = unpack(pred, mask, bg)
    pred8 = to_8b(pred[0])
    value = psnr(from_8b(pred8), from_8b(to_8b(data["target_rgbs"][0]))) if data.get("target_rgbs") is not None else None
    return pred8, value


class GraphedTrainStep:
    """The reference's training iteration (train.py:309-349: zero_grad -> forward -> unpack -> compute_loss -> backward ->
    optimizer step) captured ONCE in a HIP graph and replayed per frame.  An iteration is ~210 kernel launches of a few
    microseconds each (2.7 ms of kernels on MI355X); on a slower host than the GPU the launches, not the kernels, set the pace.
    Everything per-frame lives in static device buffers that `step(frame)` overwrites before the replay.

        step = GraphedTrainStep(model, optimizer, loss_cfg, lpips_func)    # optimizer: capturable=True
        for frame in loader:
            total = step(frame)            # frame: K, E, cnl_gtfms, dst_Rs, dst_Ts, bgcolor, target_rgbs, target_masks (+ dst_posevec)

    Python-level decisions are frozen at capture (i_iter thresholds such as kick_in_iter, loss coefficients, mesh
    topology): call `invalidate()` when one of them changes (e.g. after `model.subdivide()`), the next step re-captures."""

    FRAME_KEYS = ("K", "E", "cnl_gtfms", "dst_Rs", "dst_Ts", "dst_posevec", "bgcolor", "target_rgbs", "target_masks")

    def __init__(self, model, optimizer, loss_cfg, lpips_func=None, warmup: int = 3):
        self.model, self.opt, self.loss_cfg, self.lpips = model, optimizer, loss_cfg, lpips_func
        self.warmup, self.graph, self.static, self.total, self.i_iter = warmup, None, None, None, 0
        model.capture_safe = True
        # (experiment, off: the mesh branch and the splat rasterizer as parallel branches of the graph -- autograd replays the backward on the forward's
        #  streams.  Worth 58 us while the graph still held the camera block's ~40 tensor launches; with one camera launch it LOSES 10-25 us: 2.69 against 2.67 ms)
        self.overlap_branches = os.environ.get("GOM_GRAPH_OVERLAP_BRANCHES", "0") != "0"

    def invalidate(self):
        self.graph = None

    def _iteration(self):
        keep = getattr(self.model, "overlap_branches_train", False)
        if hasattr(self.model, "overlap_branches_train"):
            self.model.overlap_branches_train = self.overlap_branches or keep
        try:
            return self._iteration_body()
        finally:
            if hasattr(self.model, "overlap_branches_train"):
                self.model.overlap_branches_train = keep

    def _iteration_body(self):
        fr = self.static
        self.opt.zero_grad(set_to_none=True)
        if hasattr(self.lpips, "prefetch_target") and _get(self.loss_cfg, "lpips.coeff", 1.0) > 0:
            self.lpips.prefetch_target(fr["target_rgbs"])     # (a second stream inside the capture: a parallel branch of the graph)
        rgbs, masks, out = self.model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], dst_posevec=fr.get("dst_posevec"),
                                      i_iter=self.i_iter)
        total, _ = compute_loss(unpack(rgbs, masks, fr["bgcolor"]), masks, out, fr["target_rgbs"], fr["target_masks"], self.loss_cfg,
                                i_iter=self.i_iter, lpips_func=self.lpips)
        backward_from(total)
        self.opt.step()
        return total.detach()

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # allocations (library scratch, optimizer state) happen here, outside the graph
            for _ in range(self.warmup):
                first = self._iteration()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):         # (records the launches, executes nothing)
            self.total = self._iteration()
        self.total.copy_(first)

    def __call__(self, frame, i_iter=None):
        if i_iter is not None:
            self.i_iter = i_iter
        if self.static is None:
            self.static = {k: frame[k].clone() for k in self.FRAME_KEYS if k in frame and frame[k] is not None}
        else:
            for k, v in self.static.items():
                v.copy_(frame[k], non_blocking=True)
        if self.graph is None:
            self._capture()                        # (this call = `warmup` ordinary iterations on this frame, then the capture)
        else:
            self.graph.replay()
        return self.total


class GraphedRender:
    """Novel-view rendering (eval.py:334-350: eval-mode forward under no_grad, composition on the background) captured once in a
    HIP graph and replayed per frame: a frame is ~65 launches of a few microseconds, more host time than GPU time when issued one
    by one.  Same conditions as GraphedTrainStep (device-resident camera, fixed-capacity shadow pixel list: `model.capture_safe`).

        render = GraphedRender(model)
        image = render(frame)          # frame: K, E, cnl_gtfms, dst_Rs, dst_Ts, bgcolor;  returns a static (1, H, W, 3) buffer"""

    FRAME_KEYS = ("K", "E", "cnl_gtfms", "dst_Rs", "dst_Ts", "bgcolor")
    OPTIONAL_KEYS = ("dst_posevec",)      # read by the pose-refinement / non-rigid modules (eval.py:341-343 passes data['dst_posevec'])

    def __init__(self, model, warmup: int = 3, i_iter: float = 1e7):
        self.model, self.warmup, self.graph, self.static, self.out, self.i_iter = model, warmup, None, None, None, i_iter

    def invalidate(self):
        self.graph = None

    def _frame(self):
        fr = self.static
        with torch.no_grad():
            rgbs, masks, _ = self.model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], dst_posevec=fr.get("dst_posevec"), i_iter=self.i_iter)
            return unpack(rgbs, masks, fr["bgcolor"])

    def __call__(self, frame):
        if self.graph is None:
            self.model.capture_safe = True
            self.static = {k: frame[k].clone() for k in self.FRAME_KEYS}
            self.static.update({k: frame[k].clone() for k in self.OPTIONAL_KEYS if frame.get(k) is not None})
            if (self.model.pose_refinement_module is not None or self.model.non_rigid_module is not None) and "dst_posevec" not in self.static:
                raise KeyError("GraphedRender: the model has pose-conditioned modules, the frame needs 'dst_posevec'")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._frame()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._frame()
        else:
            for k in self.static:
                self.static[k].copy_(frame[k], non_blocking=True)
        self.graph.replay()
        return self.out
