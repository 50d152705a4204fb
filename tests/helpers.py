"""Shared seeded scene builders for the tests (CPU side, numpy/torch)."""
import math

import numpy as np
import torch

from gomavatar_amd import synthetic as syn
from oracle import geometry as og


def small_scene(seed=0, P=400, H=64, W=80, opacity=1.0, spread=0.6, scale=0.03, C=4, z=3.0):
    """Random anisotropic Gaussians in front of a pinhole camera.  Returns
    (cam dict, means3D, cov6, colors, opacity) as float32 numpy arrays."""
    rng = np.random.default_rng(seed)
    f = 1.2 * W
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
    E = np.eye(4, dtype=np.float32)
    # small camera rotation so that view matrix is not the identity
    E[:3, :3] = syn.rodrigues(np.array([0.05, -0.08, 0.02])).astype(np.float32)
    E[:3, 3] = np.array([0.02, -0.01, z], np.float32)
    cam = og.camera_from_KE(K, E, W, H)
    means = rng.normal(0, spread, size=(P, 3)).astype(np.float32)
    A = rng.normal(0, scale, size=(P, 3, 3)).astype(np.float32)
    cov = A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3, dtype=np.float32)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1).astype(np.float32)
    colors = rng.uniform(0, 1, size=(P, C)).astype(np.float32)
    if np.isscalar(opacity):
        op = np.full(P, opacity, np.float32)
    else:
        op = rng.uniform(opacity[0], opacity[1], size=P).astype(np.float32)
    return cam, means, cov6, colors, op


def body_scene(subdivisions=0, frame=0, img=512, body=None):
    """The synthetic GoMAvatar frame: returns dict with canonical body, params,
    frame data (torch CPU tensors) for the restated render path."""
    body = body if body is not None else syn.make_body(subdivisions)
    F = body["faces"].shape[0]
    gp = syn.make_gaussian_params(F)
    fr = {k: torch.from_numpy(v) for k, v in syn.make_frame(frame, img).items()}
    w = torch.from_numpy(body["canonical_lbs_weights"]).T
    w25 = torch.cat([w, torch.zeros(1, w.shape[1])], 0).contiguous()
    params = dict(vertices=torch.from_numpy(body["canonical_vertex"]).T.contiguous(), so3=torch.from_numpy(gp["so3"]),
                  scale=torch.from_numpy(gp["scale"]), appearance=torch.from_numpy(gp["appearance"]))
    return dict(body=body, params=params, frame=fr, lbs_weights=w25, faces=torch.from_numpy(body["faces"]), img=img)
