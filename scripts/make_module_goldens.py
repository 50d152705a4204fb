#!/usr/bin/env python
"""Records tests/golden/pose_modules.npz through the reference's OWN NonRigidModule / PoseRefinementModule classes
(/root/reference/models/modules/{non_rigid,pose_refinement}_module.py, imported here with the absent third-party names
stubbed exactly as scripts/make_goldens.py does).  Inputs come from our seeded generator; the file holds numbers only
(weights the reference's constructor drew, inputs, outputs, input gradients).  Runs in the build container only."""
import os, sys, types
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO); sys.path.insert(0, REF); sys.path.insert(0, os.path.join(REPO, "scripts"))
import make_goldens as mg  # noqa: E402

mg.install_stubs({"calls": []})
from models.modules.non_rigid_module import NonRigidModule as RefNR  # noqa: E402
from models.modules.pose_refinement_module import PoseRefinementModule as RefPR  # noqa: E402
from utils.network_util import RodriguesModule as RefRod  # noqa: E402

NS = types.SimpleNamespace
nr_cfg = NS(name="basic", condition_code_size=69, mlp_width=128, mlp_depth=6, skips=[4], multires=6, i_embed=0, kick_in_iter=150000, full_band_iter=200000)
pr_cfg = NS(name="basic", embedding_size=69, total_bones=24, mlp_width=256, mlp_depth=4, refine_root=False, refine_t=False, kick_in_iter=100000)
# parameter counts at the experiment's sizes (exps/zju-mocap_377.yaml:64-86), values at reduced widths (a small fixture)
counts = (sum(p.numel() for p in RefNR(nr_cfg).parameters()), sum(p.numel() for p in RefPR(pr_cfg).parameters()))
nr_cfg.mlp_width, pr_cfg.mlp_width = 32, 48
torch.manual_seed(11)
nr, pr = RefNR(nr_cfg), RefPR(pr_cfg)
with torch.no_grad():      # the last layers start at 1e-5: scaled up so that the outputs say something
    nr.block_mlps[-1].weight.normal_(0, 0.05); pr.block_mlps[-1].weight.normal_(0, 0.05)
g = torch.Generator().manual_seed(12)
xyz = (torch.randn(1, 3, 57, generator=g) * 0.4).requires_grad_()
pose = (torch.randn(1, 69, generator=g) * 0.3).requires_grad_()
out = {"xyz": xyz.detach().numpy(), "posevec": pose.detach().numpy()}
for it in (150000, 163000, 181000, 200000, 10000000):
    o, _, _ = nr(xyz, pose, it, R=None, S=None)
    gx, gp = torch.autograd.grad(o.square().sum(), (xyz, pose))
    out[f"nr_out_{it}"], out[f"nr_gxyz_{it}"], out[f"nr_gpose_{it}"] = o.detach().numpy(), gx.numpy(), gp.numpy()
Rs = pr(pose)
out["pr_out"] = Rs.detach().numpy()
out["pr_gpose"] = torch.autograd.grad((Rs * torch.arange(9.0).view(3, 3)).sum(), pose)[0].numpy()
out.update({"nr_" + k: v.numpy() for k, v in nr.state_dict().items()})
out.update({"pr_" + k: v.numpy() for k, v in pr.state_dict().items()})
# RodriguesModule itself (network_util.py:64-92): Model.forward's global_R branch (model.py:218-221) and train_pose.py's Rh.  Eight vectors:
# generic, large angle, and |r| at / near zero, where theta = sqrt(1e-5 + |r|^2) is NOT |r| and the "axis" r / theta is not a unit vector.
rv = torch.randn(8, 3, generator=g)
rv = (rv * torch.tensor([1.0, 0.3, 2.5, 3e-3, 1e-4, 0.0, 1e-2, 0.7])[:, None]).requires_grad_()
Rm = RefRod()(rv)
out["rod_rvec"], out["rod_out"] = rv.detach().numpy(), Rm.detach().numpy()
out["rod_grvec"] = torch.autograd.grad((Rm * torch.arange(1.0, 10.0).view(3, 3)).sum(), rv)[0].numpy()
out["nr_param_count"], out["pr_param_count"] = np.int64(counts[0]), np.int64(counts[1])
np.savez_compressed(os.path.join(REPO, "tests", "golden", "pose_modules.npz"), **out)
print("pose_modules.npz", out["nr_param_count"], out["pr_param_count"])
