#!/usr/bin/env python
"""BASELINE configs[1] at its stated shape ("3 000-iteration loop S -> (subdivide at iteration 1 000) -> M @ 512^2"), on synthetic
data, as the reference runs it: train.py:309-349 per iteration (zero_grad -> Model.forward -> unpack -> compute_loss with EVERY term of
exps/zju-mocap_377.yaml incl. LPIPS on the bf16x3 trunk -> backward -> Adam over Model.get_param_groups -> update_lr) through the drop-in
`Model`, `train_util.train_iteration` and `GomAdam`; the subdivision rebuilds the optimizer like train.py:330-346.  A teacher avatar renders the
8 target views; the student starts from the reference's initial state (scripts/train_curve_common.py -- the state scripts/train_curve_oracle.py
trains from on the CPU oracle).  Writes profiles/<tag>_train_curve.json: loss terms + 8-bit PSNR per iteration for the first --dense
iterations and every --every afterwards, mean PSNR over the 8 views at the checkpoints, iterations/s before and after the subdivision, peak memory.

    python scripts/train_synthetic.py --iters 3000 --subdivide-at 1000 [--tag r05] [--no-lpips] [--precision bf16x3]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import train_curve_common as C
from gomavatar_amd.workload import zju_cfg
from gomavatar_amd.model import Model
from gomavatar_amd import train_util as tu, metrics as M
from gomavatar_amd.lpips import LPIPSMatrixCore
from gomavatar_amd.optim import GomAdam

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000); ap.add_argument("--subdivide-at", type=int, default=1000)
ap.add_argument("--dense", type=int, default=200, help="log every iteration up to here (the span the oracle-trained curve covers)")
ap.add_argument("--every", type=int, default=50); ap.add_argument("--tag", default="r06"); ap.add_argument("--no-lpips", action="store_true")
ap.add_argument("--precision", default="bf16x3"); ap.add_argument("--out", default=None)
ap.add_argument("--save-state-at", type=int, default=-1, help="after this many iterations: parameters + Adam state -> --state-out (the start of scripts/train_curve_oracle.py --from-state)")
ap.add_argument("--state-out", default=None)
ap.add_argument("--dense-range", default="", help="A:B -- also log every iteration with A < n_iters <= B (the span an oracle run from a saved state covers)")
a = ap.parse_args()
dense_lo, dense_hi = (int(x) for x in a.dense_range.split(":")) if a.dense_range else (0, 0)
dev = "cuda"
mcfg, tcfg = zju_cfg(C.IMG, lr_decay_steps=C.LR_DECAY_STEPS)
if a.no_lpips:
    tcfg.losses.lpips.coeff = 0.0
body, tp, sp, twb, swb = C.setup(0)


def make(params, wb):
    m = Model(mcfg, body, device=dev)
    lin = [l for l in m.shadow_module.block_mlps if isinstance(l, torch.nn.Linear)]
    with torch.no_grad():
        for k in ("vertices", "so3", "scale", "appearance"):
            getattr(m, k).copy_(params[k])
        for i, l in enumerate(lin):
            l.weight.copy_(wb[2 * i].float()); l.bias.copy_(wb[2 * i + 1].float())
    return m


teacher, student = make(tp, twb).eval(), make(sp, swb).train()
frames = []
for i in range(C.N_VIEWS):
    fr = {k: torch.from_numpy(v).to(dev) for k, v in C.frame(i).items()}
    with torch.no_grad():
        rgbs, masks, _ = teacher(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
        fr["target_rgbs"], fr["target_masks"] = tu.unpack(rgbs, masks, fr["bgcolor"]).clamp(0, 1), masks.clone()   # the reference's key names (dataset/train.py:272-275)
    frames.append(fr)
lp = None if a.no_lpips else LPIPSMatrixCore(trunk_seed=0, device=dev, precision=a.precision)
opt = GomAdam(student.get_param_groups(tcfg), betas=(0.9, 0.999))


def psnr8(pred, gt):
    return float(M.psnr(M.from_8b(M.to_8b(pred)), M.from_8b(M.to_8b(gt))))


def all_views_psnr():
    was = student.training
    student.eval()
    vals = []
    with torch.no_grad():
        for fr in frames:
            rgbs, masks, _ = student(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
            vals.append(psnr8(tu.unpack(rgbs, masks, fr["bgcolor"])[0], fr["target_rgbs"][0]))
    student.train(was)
    return sum(vals) / len(vals)


log, marks, rates = [], {}, {}
torch.cuda.reset_peak_memory_stats()
torch.cuda.synchronize(); t_start = time.perf_counter()
t_seg, n_seg, seg_name = None, 0, "before_subdivision"
for it in range(a.iters):
    n_iters = it + 1                                               # train.py:268: n_iters counts from 1
    if it == a.subdivide_at:                                       # train.py:330-346: subdivide, rebuild the optimizer
        torch.cuda.synchronize()
        rates[seg_name] = round(n_seg / (time.perf_counter() - t_seg), 1)
        student.subdivide()
        opt = GomAdam(student.get_param_groups(tcfg), betas=(0.9, 0.999))
        seg_name, t_seg, n_seg = "after_subdivision", None, 0
    if t_seg is None and (it >= 20 if seg_name == "before_subdivision" else it >= a.subdivide_at + 20):   # steady state: lazy loading, allocator growth behind us
        torch.cuda.synchronize(); t_seg, n_seg = time.perf_counter(), 0
    if it == a.save_state_at:                                     # the state BEFORE iteration it + 1: what the CPU oracle continues from
        import numpy as np
        torch.cuda.synchronize()
        named = [("vertices", student.vertices), ("so3", student.so3), ("scale", student.scale), ("appearance", student.appearance)]
        named += [(f"shadow{i}", p_) for i, p_ in enumerate(l_ for lyr in student.shadow_module.block_mlps if isinstance(lyr, torch.nn.Linear) for l_ in (lyr.weight, lyr.bias))]
        opt._sync_steps() if hasattr(opt, "_sync_steps") else None
        st = {"iterations_done": np.int64(it)}
        for nm, p_ in named:
            s_ = opt.state[p_]
            st[nm] = p_.detach().cpu().numpy(); st[nm + ".exp_avg"] = s_["exp_avg"].cpu().numpy(); st[nm + ".exp_avg_sq"] = s_["exp_avg_sq"].cpu().numpy()
            st[nm + ".step"] = np.float64(float(s_["step"]))
        np.savez_compressed(a.state_out or os.path.join(ROOT, "profiles", f"{a.tag}_hip_state_iter{it}.npz"), **st)
    fr = frames[it % C.N_VIEWS]
    loss, items, rgb, mask = tu.train_iteration(student, opt, fr, tcfg, n_iters, lpips_func=lp)
    n_seg += 1
    if n_iters <= a.dense or n_iters % a.every == 0 or n_iters == a.iters or dense_lo < n_iters <= dense_hi:
        torch.cuda.synchronize()                                   # (no host read is left in an iteration: the queue the host ran ahead by is TRAINING time, not logging)
        t_log = time.perf_counter()
        row = {"iter": n_iters, "total": float(loss), "psnr": round(psnr8(rgb.detach()[0], fr["target_rgbs"][0]), 4), "faces": int(student.faces.shape[0]),
               **{k: float(v["unscaled"]) for k, v in items.items()}}
        if n_iters % (4 * a.every) == 0 or n_iters == a.iters or n_iters == a.subdivide_at:
            row["psnr_mean_8_views"] = round(all_views_psnr(), 4)
        log.append(row)
        if n_iters % a.every == 0 or n_iters == a.iters:
            print(row, flush=True)
        torch.cuda.synchronize()
        if t_seg is not None:
            t_seg += time.perf_counter() - t_log                   # (logging is not training time)
torch.cuda.synchronize()
rates[seg_name] = round(n_seg / (time.perf_counter() - t_seg), 1)
wall = time.perf_counter() - t_start
out = {"what": "BASELINE configs[1] on synthetic data through gomavatar_amd (Model + train_util.train_iteration + GomAdam + update_lr), one MI355X: "
               f"13 776 Gaussians -> subdivide at iteration {a.subdivide_at} -> 55 104, 512 x 512, 8 views, every loss term of exps/zju-mocap_377.yaml"
               + ("" if a.no_lpips else f" incl. LPIPS on the {a.precision} matrix-core trunk (seeded VGG16)"),
       "iterations": a.iters, "subdivide_at": a.subdivide_at, "wall_seconds_including_logging": round(wall, 2), "iterations_per_s": rates,
       "peak_memory_MB": round(torch.cuda.max_memory_allocated() / 2 ** 20, 1), "final_psnr_mean_8_views": log[-1].get("psnr_mean_8_views"),
       "final_psnr_last_frame": log[-1]["psnr"], "log": log}
path = a.out or os.path.join(ROOT, "profiles", f"{a.tag}_train_curve.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=0)
print(json.dumps({k: v for k, v in out.items() if k != "log"}))
