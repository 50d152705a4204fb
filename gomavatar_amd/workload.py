"""The synthetic metric workload of BASELINE.json, built once and shared by bench.py, the -m gpu parity tests
on the metric workload and the profiling scripts -- so that what is tested is what is timed.

    body      SMPL-topology UV sphere (6 890 verts / 13 776 faces), `subdiv` midpoint subdivisions (1 -> 55 104 Gaussians)
    params    seeded perturbation of the reference's initial state (seed 1); targets come from a different set (seed 2)
    frames    synthetic.make_frame(rank * 1000 + i): pose seed = frame index, the reference's orbiting synthetic camera,
              random background colour; target image / mask = the HIP render of the target parameters, composited on the
              background (train.py:53-55) -- resident in HBM, like everything else, before anything is timed."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import synthetic as syn
from .pipeline import RenderStep

POSE_KEYS = ("cnl_gtfms", "dst_Rs", "dst_Ts")


class MetricWorkload:
    def __init__(self, device, subdiv: int = 1, img: int = 512, n_frames: int = 8, rank: int = 0, first_frame: int = 0):
        dev = self.device = torch.device(device)
        self.img, self.subdiv = int(img), int(subdiv)
        body = self.body = syn.make_body(subdiv)
        self.N, self.F = body["canonical_vertex"].shape[0], body["faces"].shape[0]
        w = torch.from_numpy(body["canonical_lbs_weights"]).T
        self.w25 = torch.cat([w, torch.zeros(1, self.N)], 0).contiguous()
        self.faces = torch.from_numpy(body["faces"])
        self.params_cpu = self._params(1)
        self.target_params_cpu = self._params(2)
        self.params = {k: v.to(dev) for k, v in self.params_cpu.items()}
        self.target_params = {k: v.to(dev) for k, v in self.target_params_cpu.items()}
        self.frame_ids = [rank * 1000 + first_frame + i for i in range(n_frames)]   # each rank renders different frames
        self.frames_np = [syn.make_frame(i, img) for i in self.frame_ids]
        self.frames: List[Dict[str, torch.Tensor]] = []
        gen = RenderStep(self.faces, self.N, (img, img), self.w25, device=dev)      # single-frame instance: renders the targets
        zero_rgb, zero_m = torch.zeros((img, img, 3), device=dev), torch.zeros((img, img), device=dev)
        for fr in self.frames_np:
            d = {k: torch.from_numpy(fr[k][0]).contiguous().to(dev) for k in POSE_KEYS}
            d["K"], d["E"], d["bg"] = fr["K"][0], fr["E"][0], torch.from_numpy(fr["bgcolor"][0]).to(dev)
            gen.set_camera(d["K"], d["E"])
            gen.forward_backward(self.target_params, d, zero_rgb, zero_m, d["bg"], backward=False)
            rgb, mask = gen.rgb_mask()
            d["gt_rgb"] = (rgb[0] * mask[0, ..., None] + d["bg"] * (1 - mask[0, ..., None])).contiguous().clone()
            d["gt_mask"] = mask[0].contiguous().clone()
            self.frames.append(d)
        torch.cuda.synchronize(dev)
        del gen

    def _params(self, seed: int) -> Dict[str, torch.Tensor]:
        gp = syn.make_gaussian_params(self.F, seed)
        return dict(vertices=torch.from_numpy(self.body["canonical_vertex"]).T.contiguous(), so3=torch.from_numpy(gp["so3"]),
                    scale=torch.from_numpy(gp["scale"]), appearance=torch.from_numpy(gp["appearance"]))

    def step(self, batch: int = 1, split: int = 1):
        """split > 1: the step's `batch` frames as `split` concurrent launch sequences (pipeline.SplitRenderStep: same bits, tails filled)."""
        if not isinstance(split, int) or split > 1:
            from .pipeline import SplitRenderStep
            return SplitRenderStep(self.faces, self.N, (self.img, self.img), self.w25, device=self.device, batch=batch, split=split)
        return RenderStep(self.faces, self.N, (self.img, self.img), self.w25, device=self.device, batch=batch)

    def batches(self, step: RenderStep) -> List[Dict[str, torch.Tensor]]:
        """Consecutive groups of `step.B` frames: stacked per-frame inputs + the device camera array of each group."""
        B = step.B
        out = []
        for j in range(len(self.frames) // B):
            grp = self.frames[j * B:(j + 1) * B]
            bt = {k: torch.stack([g[k] for g in grp]).contiguous() for k in POSE_KEYS + ("gt_rgb", "gt_mask", "bg")}
            if B == 1:
                bt = {k: v[0] for k, v in bt.items()}
                step.set_camera(grp[0]["K"], grp[0]["E"])
            else:
                step.set_cameras([g["K"] for g in grp], [g["E"] for g in grp])
            torch.cuda.synchronize(self.device)
            bt["cams_dev"], bt["cam"], bt["frames"] = step.cams_dev.clone(), step.cam, list(range(j * B, (j + 1) * B))
            out.append(bt)
        return out

    def oracle_frame(self, i: int) -> Dict[str, torch.Tensor]:
        """Frame i as the CPU oracle's `render_path` takes it (batch dimension 1, host tensors)."""
        return {k: torch.from_numpy(v) for k, v in self.frames_np[i].items()}


# ---- BASELINE configs[1] / configs[3] as the reference runs them: the `Model` iteration on the metric workload ------------------------------
def zju_cfg(img: int = 512, non_rigid_kick_in: int = 150000, pose_kick_in: int = 100000, lr_decay_steps: float = 100000):
    """(model cfg, train cfg) with the values of configs/default.yaml + exps/zju-mocap_377.yaml that the hot path reads, as attribute nodes."""
    from types import SimpleNamespace as NS
    mcfg = NS(img_size=(img, img), canonical_geometry=NS(sigma=1e-3, radius_scale=1.0, deform_so3=True, deform_scale=True), appearance=NS(color_init=0.5),
              normal_renderer=NS(sigma=1e-5, soft_mask=True), shadow_module=NS(name="basic", multires=6, mlp_width=128, mlp_depth=3, skips=(4,)),
              lbs_weights=NS(refine=False),
              non_rigid=NS(name="basic", condition_code_size=69, mlp_width=128, mlp_depth=6, skips=[4], multires=6, i_embed=0, kick_in_iter=non_rigid_kick_in,
                           full_band_iter=non_rigid_kick_in + 50000),
              pose_refinement=NS(name="basic", embedding_size=69, total_bones=24, mlp_width=256, mlp_depth=4, refine_root=False, refine_t=False, kick_in_iter=pose_kick_in))
    tcfg = NS(lr=NS(lbs_weights=0.0, appearance=0.0005, canonical_geometry=0.0005, canonical_geometry_xyz=0.0005, non_rigid=0.0005, pose_refinement=0.00005, shadow=0.0005),
              lr_decay_steps=lr_decay_steps,
              losses=NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=1.0), laplacian=NS(coeff_canonical=0.0, coeff_observation=10.0),
                        normal=NS(coeff_mask=1.0, kernel_size=7, coeff_consist=0.1), color_consist=NS(coeff=0.05)))
    return mcfg, tcfg


def model_frames(wl: "MetricWorkload", ids=None):
    """The workload's frames as the reference's data dicts (dataset/train.py:209-287: batch dimension 1, `target_rgbs` / `target_masks`)."""
    out = []
    for i in (range(len(wl.frames)) if ids is None else ids):
        fr = {k: torch.from_numpy(v).to(wl.device) for k, v in wl.frames_np[i].items()}
        fr["target_rgbs"], fr["target_masks"] = wl.frames[i]["gt_rgb"][None], wl.frames[i]["gt_mask"][None]
        out.append(fr)
    return out


def build_model(wl: "MetricWorkload", mcfg, with_mlps: bool = True, seed: int = 0):
    """The drop-in `Model` on the workload's body, with the reference's two optional MLPs (modules.NonRigidModule / PoseRefinementModule:
    334 152 of the 951 023 parameters of SURVEY.md 8(e)) when `with_mlps`.  Seeded: every rank builds the same weights."""
    from .model import Model
    from .modules import NonRigidModule, PoseRefinementModule
    torch.manual_seed(seed)
    nr = NonRigidModule(mcfg.non_rigid).to(wl.device) if with_mlps else None
    pr = PoseRefinementModule(mcfg.pose_refinement).to(wl.device) if with_mlps else None
    return Model(mcfg, wl.body, non_rigid_module=nr, pose_refinement_module=pr, device=wl.device).train()
