"""What the two training-curve scripts share (scripts/train_synthetic.py: the HIP path; scripts/train_curve_oracle.py: the CPU oracle): BASELINE
configs[1] on synthetic data -- body, teacher, the student's initial state, frames, coefficients -- so that "PSNR vs iteration, HIP vs
oracle-trained"  compares two runs of the SAME problem from the SAME initialisation.  No oracle import, no HIP import."""
import numpy as np
import torch

from gomavatar_amd import synthetic as syn

IMG, N_VIEWS = 512, 8
LR = dict(lbs_weights=0.0, appearance=0.0005, canonical_geometry=0.0005, canonical_geometry_xyz=0.0005, non_rigid=0.0005, pose_refinement=0.00005, shadow=0.0005)   # exps/zju-mocap_377.yaml:113-119
LR_DECAY_STEPS = 100000


def shadow_weights(seed: int, last_std: float):
    """[W1, b1, ..., W4, b4] of the shadow MLP (39 -> 128 -> 128 -> 128 -> 1): Xavier-uniform like network_util.initseq; the last layer N(0, last_std)
    (the reference starts it at U(-1e-5, 1e-5): the student uses 1e-5, the teacher 0.3 -- a shading that really varies)."""
    g = torch.Generator().manual_seed(seed)
    wb = []
    for i, (o, n) in enumerate([(128, 39), (128, 128), (128, 128), (1, 128)]):
        bound = (6.0 / (o + n)) ** 0.5 * (2.0 ** 0.5 if i < 3 else 1.0)
        w = (torch.rand(o, n, generator=g, dtype=torch.float64) * 2 - 1) * bound if i < 3 else torch.randn(o, n, generator=g, dtype=torch.float64) * last_std
        wb += [w, torch.zeros(o, dtype=torch.float64)]
    return wb


def setup(level: int = 0):
    body = syn.make_body(level)
    F = body["faces"].shape[0]
    verts = torch.from_numpy(body["canonical_vertex"]).T.contiguous()
    gp = syn.make_gaussian_params(F, 2)
    teacher = dict(vertices=verts.clone(), so3=torch.from_numpy(gp["so3"]), scale=torch.from_numpy(gp["scale"]), appearance=torch.from_numpy(gp["appearance"]))
    # the reference's initial state (models/model.py:74-85, appearance_module.py:14): zero rotations, unit scales, grey
    student = dict(vertices=verts.clone(), so3=torch.zeros(3, F), scale=torch.ones(3, F), appearance=torch.full((3, F), 0.5))
    return body, teacher, student, shadow_weights(3, 0.3), shadow_weights(4, 1e-5)


def frame(i: int):
    return syn.make_frame(i % N_VIEWS, IMG)
