"""Seeded synthetic stand-ins for the assets the reference loads from disk.

The reference trains on ZJU-MoCap / PeopleSnapshot with a licensed SMPL
template; none of that is available offline.  This module produces inputs of
exactly the same *shape and schema* (SURVEY.md section 8d):

* a closed genus-0 body mesh with SMPL's counts (6 890 verts / 13 776 faces),
* the 24-joint SMPL kinematic tree (parent table: reference
  ``utils/body_util.py:36-39``) with a T-pose joint table,
* top-4 distance-softmax LBS weights, stored as the reference's
  ``canonical_info['canonical_lbs_weights']`` (N, 24) array,
* midpoint subdivision with the reference's child-face order
  (``utils/pc_util.py:49-163``: children of face f are 4f..4f+3 =
  (v0,m0,m2) (m0,v1,m1) (m2,m1,v2) (m0,m1,m2)),
* the per-frame data dict of ``dataset/train.py:209-287``
  (K, E, cnl_gtfms, dst_Rs, dst_Ts, dst_posevec, bgcolor, ...), using the
  reference's own synthetic camera (``dataset/newpose.py:33-36,86-104``).

numpy only; everything is a pure function of integer seeds.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

# reference utils/body_util.py:36-39
SMPL_PARENT = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)

# metres, pelvis at the origin, y up, T-pose.  Hand-made table (not SMPL data).
TPOSE_JOINTS = np.array(
    [
        [0.00, 0.00, 0.00],    # 0 pelvis
        [0.07, -0.09, 0.00],   # 1 l_hip
        [-0.07, -0.09, 0.00],  # 2 r_hip
        [0.00, 0.11, -0.01],   # 3 spine1
        [0.10, -0.47, 0.00],   # 4 l_knee
        [-0.10, -0.47, 0.00],  # 5 r_knee
        [0.00, 0.25, 0.00],    # 6 spine2
        [0.09, -0.87, -0.03],  # 7 l_ankle
        [-0.09, -0.87, -0.03], # 8 r_ankle
        [0.00, 0.31, 0.01],    # 9 spine3
        [0.12, -0.93, 0.09],   # 10 l_foot
        [-0.12, -0.93, 0.09],  # 11 r_foot
        [0.00, 0.52, -0.03],   # 12 neck
        [0.08, 0.43, -0.02],   # 13 l_collar
        [-0.08, 0.43, -0.02],  # 14 r_collar
        [0.00, 0.60, 0.02],    # 15 head
        [0.18, 0.46, -0.02],   # 16 l_shoulder
        [-0.18, 0.46, -0.02],  # 17 r_shoulder
        [0.44, 0.45, -0.03],   # 18 l_elbow
        [-0.44, 0.45, -0.03],  # 19 r_elbow
        [0.69, 0.45, -0.02],   # 20 l_wrist
        [-0.69, 0.45, -0.02],  # 21 r_wrist
        [0.78, 0.44, -0.02],   # 22 l_hand
        [-0.78, 0.44, -0.02],  # 23 r_hand
    ],
    dtype=np.float32,
)

BODY_RADII = (0.35, 0.875, 0.22)  # x, y, z semi-axes of the body ellipsoid (m)
BODY_CENTER_Y = -0.075            # pelvis is a little above the ellipsoid centre


def uv_sphere(rings: int = 82, segments: int = 84) -> Tuple[np.ndarray, np.ndarray]:
    """Unit UV sphere: rings*segments + 2 verts, 2*segments*rings faces.

    82 x 84 gives exactly SMPL's 6 890 vertices and 13 776 faces.
    Faces are wound counter-clockwise seen from outside.
    """
    verts = [(0.0, 1.0, 0.0)]
    for r in range(rings):
        phi = math.pi * (r + 1) / (rings + 1)
        for s in range(segments):
            th = 2.0 * math.pi * s / segments
            verts.append((math.sin(phi) * math.cos(th), math.cos(phi), math.sin(phi) * math.sin(th)))
    verts.append((0.0, -1.0, 0.0))
    verts = np.asarray(verts, dtype=np.float64)
    south = len(verts) - 1

    def vid(r, s):
        return 1 + r * segments + (s % segments)

    faces = []
    for s in range(segments):
        faces.append((0, vid(0, s + 1), vid(0, s)))
    for r in range(rings - 1):
        for s in range(segments):
            a, b = vid(r, s), vid(r, s + 1)
            c, d = vid(r + 1, s), vid(r + 1, s + 1)
            faces.append((a, b, d))
            faces.append((a, d, c))
    for s in range(segments):
        faces.append((south, vid(rings - 1, s), vid(rings - 1, s + 1)))
    return verts, np.asarray(faces, dtype=np.int64)


def _point_segment_dist(p: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    ab = b - a
    den = float(ab @ ab)
    if den < 1e-12:
        return np.linalg.norm(p - a, axis=1)
    t = np.clip(((p - a) @ ab) / den, 0.0, 1.0)
    return np.linalg.norm(p - (a + t[:, None] * ab), axis=1)


def make_lbs_weights(verts: np.ndarray, joints: np.ndarray, sigma: float = 0.05, topk: int = 4) -> np.ndarray:
    """(N, 24) weights: softmax(-d^2 / 2 sigma^2) of distance to each joint's
    bone (segment parent->joint; the pelvis uses the point), top-k kept."""
    n = verts.shape[0]
    d = np.empty((n, 24), dtype=np.float64)
    for j in range(24):
        pj = SMPL_PARENT[j]
        if pj < 0:
            d[:, j] = np.linalg.norm(verts - joints[j], axis=1)
        else:
            d[:, j] = _point_segment_dist(verts, joints[pj].astype(np.float64), joints[j].astype(np.float64))
    logit = -(d ** 2) / (2.0 * sigma * sigma)
    logit -= logit.max(axis=1, keepdims=True)
    w = np.exp(logit)
    keep = np.argsort(-w, axis=1, kind="stable")[:, :topk]   # exactly top-k (ties broken by joint index)
    mask = np.zeros_like(w, dtype=bool)
    np.put_along_axis(mask, keep, True, axis=1)
    w = np.where(mask, w, 0.0)
    w /= w.sum(axis=1, keepdims=True)
    return w.astype(np.float32)


def subdivide(verts: np.ndarray, faces: np.ndarray, attrs: Dict[str, np.ndarray] | None = None):
    """One midpoint subdivision: V <- V+E, F <- 4F.  Per-vertex attributes are
    averaged onto the new midpoints (reference utils/pc_util.py:139-152,
    generic branch).  Children of face f are rows 4f..4f+3."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], axis=1).reshape(-1, 2)
    e = np.sort(e, axis=1)
    uniq, inverse = np.unique(e, axis=0, return_inverse=True)
    inverse = np.asarray(inverse).reshape(-1)
    mid = verts[uniq].mean(axis=1)
    m = inverse.reshape(-1, 3) + len(verts)
    f = np.column_stack(
        [faces[:, 0], m[:, 0], m[:, 2], m[:, 0], faces[:, 1], m[:, 1], m[:, 2], m[:, 1], faces[:, 2], m[:, 0], m[:, 1], m[:, 2]]
    ).reshape(-1, 3)
    new_verts = np.vstack([verts, mid])
    out_attrs = None
    if attrs is not None:
        out_attrs = {k: np.vstack([v, v[uniq].mean(axis=1)]) for k, v in attrs.items()}
    return new_verts, f.astype(np.int64), out_attrs


def make_body(subdivisions: int = 0, seed: int = 0) -> Dict[str, np.ndarray]:
    """canonical_info dict as the reference's Model.__init__ consumes it
    (models/model.py:45-85): canonical_vertex (N,3), faces (F,3),
    canonical_lbs_weights (N,24), canonical_joints (24,3)."""
    v, f = uv_sphere()
    rng = np.random.default_rng(seed)
    v = v * np.asarray(BODY_RADII)[None, :]
    v[:, 1] += BODY_CENTER_Y
    v = v + rng.normal(0.0, 1e-3, size=v.shape)  # 1 mm jitter
    w = make_lbs_weights(v, TPOSE_JOINTS)
    attrs = {"weights": w.astype(np.float64)}
    for _ in range(subdivisions):
        v, f, attrs = subdivide(v, f, attrs)
    return {
        "canonical_vertex": v.astype(np.float32),
        "faces": f,
        "canonical_lbs_weights": attrs["weights"].astype(np.float32),
        "canonical_joints": TPOSE_JOINTS.copy(),
    }


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """Axis-angle -> rotation, same formula as reference
    utils/body_util.py:288-307 (note the +1e-5 in the normalisation)."""
    rvec = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = np.linalg.norm(rvec)
    r = rvec / (theta + 1e-5)
    k = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return math.cos(theta) * np.eye(3) + math.sin(theta) * k + (1 - math.cos(theta)) * np.outer(r, r)


def pose_to_body_RTs(pose72: np.ndarray, joints: np.ndarray):
    """Local per-joint (R, T) as reference body_pose_to_body_RTs
    (utils/body_util.py:332-363)."""
    pose = pose72.reshape(24, 3)
    Rs = np.zeros((24, 3, 3), dtype=np.float32)
    Ts = np.zeros((24, 3), dtype=np.float32)
    for i in range(24):
        Rs[i] = rodrigues(pose[i])
        Ts[i] = joints[i] if i == 0 else joints[i] - joints[SMPL_PARENT[i]]
    return Rs, Ts


def canonical_global_tfms(joints: np.ndarray) -> np.ndarray:
    """(24,4,4) canonical (identity-rotation) global transforms, as reference
    get_canonical_global_tfms (utils/body_util.py:400-424)."""
    g = np.zeros((24, 4, 4), dtype=np.float32)
    for i in range(24):
        loc = np.eye(4, dtype=np.float32)
        loc[:3, 3] = joints[i] if i == 0 else joints[i] - joints[SMPL_PARENT[i]]
        g[i] = loc if i == 0 else g[SMPL_PARENT[i]] @ loc
    return g


def random_pose(seed: int, std: float = 0.25, clip: float = 0.8) -> np.ndarray:
    rng = np.random.default_rng(1000 + seed)
    pose = np.clip(rng.normal(0.0, std, size=72), -clip, clip).astype(np.float32)
    pose[:3] = 0.0
    return pose


def look_at_camera(img_size: int, yaw: float = 0.0, radius: float = 8.0, focal: float | None = None, target_y: float = BODY_CENTER_Y):
    """The reference's synthetic camera (dataset/newpose.py:86-104 +
    utils/camera_util.py:52-80 with inv_camera=True), orbiting by `yaw`.
    focal defaults to 1250 px at 512 (scaled with resolution)."""
    if focal is None:
        focal = 1250.0 * img_size / 512.0
    campos = np.array([radius * math.sin(yaw), target_y, radius * math.cos(yaw)], dtype=np.float64)
    lookat = np.array([0.0, target_y, 0.0])
    up = np.array([0.0, -1.0, 0.0])
    fwd = lookat - campos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    up2 = np.cross(fwd, right)
    up2 /= np.linalg.norm(up2)
    rot = np.stack([right, up2, fwd]).astype(np.float32)
    E = np.eye(4, dtype=np.float32)
    E[:3, :3] = rot
    E[:3, 3] = -rot @ campos.astype(np.float32)
    K = np.eye(3, dtype=np.float32)
    K[0, 0] = K[1, 1] = focal
    K[:2, 2] = img_size / 2.0
    return K, E


def make_frame(frame: int, img_size: int = 512, joints: np.ndarray = TPOSE_JOINTS, n_views: int = 8) -> Dict[str, np.ndarray]:
    """One per-frame data dict (batch dim 1), keys as dataset/train.py:209-287."""
    pose = random_pose(frame)
    Rs, Ts = pose_to_body_RTs(pose, joints)
    K, E = look_at_camera(img_size, yaw=2.0 * math.pi * (frame % n_views) / n_views)
    rng = np.random.default_rng(2000 + frame)
    return {
        "K": K[None],
        "E": E[None],
        "cnl_gtfms": canonical_global_tfms(joints)[None],
        "dst_Rs": Rs[None],
        "dst_Ts": Ts[None],
        "dst_posevec": (pose[3:] + 1e-2)[None].astype(np.float32),
        "bgcolor": rng.uniform(0.0, 1.0, size=(1, 3)).astype(np.float32),
    }


def make_gaussian_params(n_faces: int, seed: int = 1) -> Dict[str, np.ndarray]:
    """Per-face learnables in the reference's channel-first layout
    (models/model.py:74-85, appearance_module.py:14): so3 (3,F), scale (3,F),
    appearance (3,F).  Perturbed around the reference's init (0 / 1 / 0.5)."""
    rng = np.random.default_rng(seed)
    return {
        "so3": rng.normal(0.0, 0.1, size=(3, n_faces)).astype(np.float32),
        "scale": (1.0 + rng.normal(0.0, 0.05, size=(3, n_faces))).astype(np.float32),
        "appearance": rng.uniform(0.0, 1.0, size=(3, n_faces)).astype(np.float32),
    }


def icosphere_body(level: int = 2, seed: int = 0) -> Dict[str, np.ndarray]:
    """Small closed mesh (level 2: 162 verts / 320 faces) for fast tests."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]],
        dtype=np.float64,
    )
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]],
        dtype=np.int64,
    )
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(level):
        v, f, _ = subdivide(v, f, None)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    rng = np.random.default_rng(seed)
    v = v * np.asarray(BODY_RADII)[None, :]
    v[:, 1] += BODY_CENTER_Y
    v = v + rng.normal(0.0, 1e-3, size=v.shape)
    return {
        "canonical_vertex": v.astype(np.float32),
        "faces": f,
        "canonical_lbs_weights": make_lbs_weights(v, TPOSE_JOINTS),
        "canonical_joints": TPOSE_JOINTS.copy(),
    }
