"""gomavatar_amd.modules (the optional non-rigid / pose-refinement MLPs in front of the hot path) against tests/golden/pose_modules.npz,
recorded through the reference's own classes (scripts/make_module_goldens.py): same state-dict keys, same outputs and input gradients
at five points of the positional encoding's fade-in window, and the parameter counts SURVEY.md 8(e) adds up to 951 023."""
import os
from types import SimpleNamespace as NS

import numpy as np
import torch

from gomavatar_amd.modules import NonRigidModule, PoseRefinementModule

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_modules.npz"), allow_pickle=False)


def _cfgs(wn=128, wp=256):
    return (NS(name="basic", condition_code_size=69, mlp_width=wn, mlp_depth=6, skips=[4], multires=6, i_embed=0, kick_in_iter=150000, full_band_iter=200000),
            NS(name="basic", embedding_size=69, total_bones=24, mlp_width=wp, mlp_depth=4, refine_root=False, refine_t=False, kick_in_iter=100000))


def test_parameter_counts_and_init():
    nr, pr = (cls(c) for cls, c in zip((NonRigidModule, PoseRefinementModule), _cfgs()))
    assert sum(p.numel() for p in nr.parameters()) == int(G["nr_param_count"]) == 101123
    assert sum(p.numel() for p in pr.parameters()) == int(G["pr_param_count"]) == 233029
    for m in (nr, pr):      # the offsets / corrections start at (almost) nothing
        assert float(m.block_mlps[-1].weight.detach().abs().max()) <= 1e-5 and float(m.block_mlps[-1].bias.detach().abs().max()) == 0.0


def _load(m, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(prefix) and (k.endswith(".weight") or k.endswith(".bias"))}
    assert set(sd) == set(m.state_dict()), "state-dict keys differ from the reference's"
    m.load_state_dict(sd)


def test_non_rigid_matches_reference():
    nr = NonRigidModule(_cfgs(32, 48)[0])
    _load(nr, "nr_")
    for it in (150000, 163000, 181000, 200000, 10000000):
        xyz, pose = torch.from_numpy(G["xyz"]).requires_grad_(), torch.from_numpy(G["posevec"]).requires_grad_()
        o, R, S = nr(xyz, pose, it, R=None, S=None)
        assert R is None and S is None
        gx, gp = torch.autograd.grad(o.square().sum(), (xyz, pose))
        np.testing.assert_allclose(o.detach().numpy(), G[f"nr_out_{it}"], rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(gx.numpy(), G[f"nr_gxyz_{it}"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(gp.numpy(), G[f"nr_gpose_{it}"], rtol=2e-5, atol=2e-6)


def test_pose_refinement_matches_reference():
    pr = PoseRefinementModule(_cfgs(32, 48)[1])
    _load(pr, "pr_")
    pose = torch.from_numpy(G["posevec"]).requires_grad_()
    Rs = pr(pose)
    assert Rs.shape == (1, 24, 3, 3) and torch.equal(Rs[0, 0], torch.eye(3))
    np.testing.assert_allclose(Rs.detach().numpy(), G["pr_out"], rtol=2e-6, atol=2e-7)
    g = torch.autograd.grad((Rs * torch.arange(9.0).view(3, 3)).sum(), pose)[0]
    np.testing.assert_allclose(g.numpy(), G["pr_gpose"], rtol=2e-5, atol=2e-6)


def test_rodrigues_matches_reference_module():
    """modules.rodrigues == utils/network_util.py:64-92 RodriguesModule (Model.forward's global_R branch, model.py:218-221; train_pose.py's Rh):
    eight vectors incl. |r| = 0 and |r| ~ 1e-4, where sqrt(1e-5 + |r|^2) differs from |r| and the result is NOT a rotation about a unit axis."""
    from gomavatar_amd.modules import rodrigues
    rv = torch.from_numpy(G["rod_rvec"]).requires_grad_()
    R = rodrigues(rv)
    np.testing.assert_allclose(R.detach().numpy(), G["rod_out"], rtol=0, atol=1e-7)
    g = torch.autograd.grad((R * torch.arange(1.0, 10.0).view(3, 3)).sum(), rv)[0]
    np.testing.assert_allclose(g.numpy(), G["rod_grvec"], rtol=1e-5, atol=2e-6)     # (x * x against x ** 2: another autograd formula, same function)
    # the unit-axis formula the branch used before round 6 is measurably something else near zero (this is what the golden pins)
    th = rv.detach().norm(dim=1).clamp_min(1e-8)
    assert float((th - torch.sqrt(1e-5 + th * th)).abs().max()) > 1e-3
