"""On-disk formats of the reference, so that its checkpoints and prepared datasets load unchanged.

Checkpoints (train.py:269-295,370-377): `torch.save({'iter', 'network': model.state_dict(), 'optimizer': ...})`; a run that
subdivided the mesh is resumed by subdividing as often as `cfg.model.subdivide_iters` says before loading.  The only
name that differs from `gomavatar_amd.model.Model` is the colour parameter (`appearance_module.appearance` /
`appearance_module.bg_col` in the reference).

Datasets (dataset/train.py:75-126,209-287; scripts/prepare_zju-mocap/prepare_dataset.py): `cameras.pkl`
{frame: {'intrinsics', 'extrinsics'[, 'distortions']}}, `mesh_infos.pkl` {frame: {'Rh', 'Th', 'poses', 'joints',
'tpose_joints'}}, `canonical_joints.pkl` {'joints', 'vertex', 'weights'[, 'edges', 'faces']}, `images/*.png`, `masks/*.png`."""
from __future__ import annotations

import os
import pickle
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from . import synthetic as _syn

_RENAME = {"appearance_module.appearance": "appearance", "appearance_module.bg_col": "bg_col"}
_RENAME_BACK = {v: k for k, v in _RENAME.items()}


def to_reference_state_dict(model) -> Dict[str, torch.Tensor]:
    """`model.state_dict()` with the reference's key names (plus the buffers it registers: faces, target_edge_length)."""
    sd = {}
    for k, v in model.state_dict().items():
        if k.startswith("normal_renderer."):
            continue
        sd[_RENAME_BACK.get(k, k)] = v.detach().clone()
    sd["faces"] = model.faces.detach().clone()
    sd["target_edge_length"] = model.target_edge_length.detach().clone()
    return sd


def load_reference_state_dict(model, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
    """Load a reference `network` state dict; subdivides the model first when the checkpoint's mesh is a subdivision of it."""
    n_faces = int(sd["faces"].shape[0]) if "faces" in sd else int(sd["so3"].shape[1])
    if "faces" in sd and "vertices" in sd and not (model.faces.shape == sd["faces"].shape and torch.equal(model.faces.cpu(), sd["faces"].cpu().to(model.faces.dtype))):
        # The checkpoint carries its own (possibly subdivided) topology: adopt it instead of re-deriving it.  The reference numbers
        # the midpoints of a subdivision in trimesh's `grouping.unique_rows` order (utils/pc_util.py:95-121), which this package's
        # `synthetic.subdivide` (np.unique, lexicographic) need not reproduce -- the same surface with another vertex / face
        # numbering -- so a re-subdivided model would only match by luck.  Everything per-vertex / per-face is in the state dict.
        _adopt_topology(model, sd)
    else:
        while model.faces.shape[0] < n_faces:
            model.subdivide()
    if model.faces.shape[0] != n_faces:
        raise ValueError(f"checkpoint has {n_faces} faces, the model {model.faces.shape[0]} (not a midpoint subdivision of each other)")
    own = dict(model.named_parameters())
    own.update(dict(model.named_buffers()))
    used = set()
    with torch.no_grad():
        for k, v in sd.items():
            name = _RENAME.get(k, k)
            if name in ("faces", "target_edge_length"):
                if name == "faces" and not torch.equal(model.faces.cpu(), v.cpu().to(model.faces.dtype)):
                    raise ValueError("checkpoint faces differ from the subdivided canonical mesh")
                used.add(name)
                continue
            if name == "lbs_weights":
                model.lbs_weights = v.to(model.vertices.device, torch.float32).contiguous()
                used.add(name)
                continue
            if name not in own:
                if strict:
                    raise KeyError(f"unexpected key in checkpoint: {k}")
                continue
            own[name].copy_(v.to(own[name].device, own[name].dtype))
            used.add(name)
    missing = [k for k in own if k not in used and not k.startswith("normal_renderer.")]
    if strict and missing:
        raise KeyError(f"keys missing from checkpoint: {missing}")
    model._rebuild_topology()
    if "target_edge_length" in sd:   # a buffer of the reference model: the edge lengths at (the last) subdivision, not today's
        model.target_edge_length = sd["target_edge_length"].to(model.vertices.device, torch.float32)


def _adopt_topology(model, sd: Dict[str, torch.Tensor]) -> None:
    """Resize the model to the mesh of a checkpoint: faces from the checkpoint, per-vertex / per-face parameters re-created with
    the checkpoint's shapes (the vertices with the checkpoint's values, so that the edge lengths `_rebuild_topology` derives are real
    even when the caller loads non-strictly; the rest is copied by the caller), adjacency rebuilt.  The model's nn.Parameters are
    REPLACED: an optimizer built before the load still points at the old ones and must be rebuilt (as train.py:341-346 does after a
    subdivision)."""
    import torch.nn as nn
    dev = model.vertices.device
    faces = sd["faces"].to(dev, model.faces.dtype).contiguous()
    N, F = int(sd["vertices"].shape[1]), int(faces.shape[0])
    if int(faces.max()) >= N or int(faces.min()) < 0:
        raise ValueError("checkpoint faces index vertices the checkpoint does not have")
    model.faces = faces
    model.vertices = nn.Parameter(sd["vertices"].to(dev, torch.float32).contiguous().clone())
    for name in ("so3", "scale", "appearance"):
        old = getattr(model, name)
        setattr(model, name, nn.Parameter(torch.zeros(3, F, device=dev), requires_grad=old.requires_grad))
    if "lbs_weights" in sd:
        model.lbs_weights = sd["lbs_weights"].to(dev, torch.float32).contiguous()
    elif model.lbs_weights.shape[1] != N:
        raise KeyError("checkpoint with its own topology but without lbs_weights")
    model._rebuild_topology()


def save_checkpoint(path: str, n_iter: int, model, optimizer) -> None:
    """train.py:370-377."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"iter": n_iter, "network": to_reference_state_dict(model), "optimizer": optimizer.state_dict()}, path)


def load_checkpoint(path: str, model, optimizer=None, subdivide_iters: Iterable[int] = ()) -> int:
    """train.py:276-291: returns the iteration to continue from."""
    ckpt = torch.load(path, map_location="cpu")
    max_iter = int(ckpt["iter"])
    for i in subdivide_iters:
        if max_iter >= i:
            model.subdivide()
    load_reference_state_dict(model, ckpt["network"])
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    return max_iter + 1


# ---------------------------------------------------------------------------------------------------------------------
def apply_global_tfm_to_camera(E: np.ndarray, Rh: np.ndarray, Th: np.ndarray):
    """utils/camera_util.py:111-131: fold the body's global rotation / translation into the extrinsics."""
    g = np.eye(4)
    rot = _syn.rodrigues(np.asarray(Rh, dtype=np.float64).reshape(3)).T
    g[:3, :3] = rot
    g[:3, 3] = -rot.dot(np.asarray(Th, dtype=np.float64).reshape(3))
    return E.dot(np.linalg.inv(g)), g


class ReferenceDataset:
    """Frames of a dataset directory prepared by the reference's scripts, as the dict `dataset/train.py:209-287` returns: lens
    undistortion, resizing (`target_size` = (w, h) or `resize_img_scale` = (sx, sy)) and the random crop (`crop_size` = (w, h)) as
    `dataset/train.py:138-200` does them, through the numpy restatements of the OpenCV calls in `imageops.py` (OpenCV absent: parity
    unpinned).  Images need Pillow and are optional."""

    def __init__(self, dataset_path: str, bgcolor=None, load_images: bool = True, target_size=None, resize_img_scale=(1.0, 1.0),
                 crop_size=(-1, -1)):
        self.path = dataset_path
        self.target_size = tuple(target_size) if target_size is not None else None
        self.resize_img_scale = (float(resize_img_scale[0]), float(resize_img_scale[1])) if np.ndim(resize_img_scale) else (float(resize_img_scale),) * 2
        self.crop_size = tuple(crop_size)
        with open(os.path.join(dataset_path, "canonical_joints.pkl"), "rb") as f:
            cj = pickle.load(f)
        self.canonical_joints = cj["joints"].astype("float32")
        self.canonical_vertex = cj["vertex"].astype("float32")
        self.canonical_lbs_weights = cj["weights"].astype("float32")
        self.faces = cj.get("faces")
        self.edges = cj["edges"].astype(int) if "edges" in cj else None
        with open(os.path.join(dataset_path, "cameras.pkl"), "rb") as f:
            self.cameras = pickle.load(f)
        with open(os.path.join(dataset_path, "mesh_infos.pkl"), "rb") as f:
            self.mesh_infos = pickle.load(f)
        img_dir = os.path.join(dataset_path, "images")
        if os.path.isdir(img_dir):
            self.framelist = sorted(os.path.splitext(n)[0] for n in os.listdir(img_dir) if n.endswith(".png"))
        else:
            self.framelist = sorted(self.mesh_infos.keys())
        self.bgcolor = bgcolor
        self.load_images = load_images and os.path.isdir(img_dir)
        if not self.load_images and (self.target_size is not None or self.crop_size != (-1, -1)):
            # dataset/train.py:158-163,255-256 derive both from the loaded image's size: without images the intrinsics cannot follow
            raise ValueError("target_size / crop_size need the images (their scale and offset come from the image size)")

    def get_canonical_info(self):
        return {"canonical_joints": self.canonical_joints, "canonical_vertex": self.canonical_vertex,
                "canonical_lbs_weights": self.canonical_lbs_weights, "edges": self.edges, "faces": self.faces}

    def __len__(self):
        return len(self.framelist)

    def __getitem__(self, idx: int) -> Dict[str, np.ndarray]:
        name = self.framelist[idx]
        mi, cam = self.mesh_infos[name], self.cameras[name]
        bg = (np.random.rand(3) * 255.0).astype("float32") if self.bgcolor is None else np.asarray(self.bgcolor, dtype="float32")
        poses, tpose = mi["poses"].astype("float32"), mi["tpose_joints"].astype("float32")
        E, gt = apply_global_tfm_to_camera(cam["extrinsics"], mi["Rh"].astype("float32"), mi["Th"].astype("float32"))
        dst_Rs, dst_Ts = _syn.pose_to_body_RTs(poses.reshape(-1), tpose)
        K = cam["intrinsics"][:3, :3].astype(np.float64).copy()
        out = {"frame_name": name, "bgcolor": bg / 255.0, "K": K.astype(np.float32), "E": E.astype(np.float32),
               "global_tfms": gt, "dst_poses": poses, "dst_Rs": dst_Rs, "dst_Ts": dst_Ts,
               "cnl_gtfms": _syn.canonical_global_tfms(self.canonical_joints), "dst_posevec": poses.reshape(-1)[3:] + 1e-2,
               "dst_tpose_joints": tpose}
        if not self.load_images and self.resize_img_scale != (1.0, 1.0):   # dataset/train.py:239-244: K follows the resize whether or not the pixels are read
            K[:1] *= self.resize_img_scale[0]
            K[1:2] *= self.resize_img_scale[1]
            out["K"] = K.astype(np.float32)
        if self.load_images:
            from PIL import Image
            from . import imageops as iop
            img8 = np.asarray(Image.open(os.path.join(self.path, "images", name + ".png")).convert("RGB"))
            mask8 = np.asarray(Image.open(os.path.join(self.path, "masks", name + ".png")).convert("RGB"))
            orig_H, orig_W = img8.shape[:2]
            if "distortions" in cam:                                      # dataset/train.py:151-155
                img8 = iop.undistort(img8, cam["intrinsics"], cam["distortions"])
                mask8 = iop.undistort(mask8, cam["intrinsics"], cam["distortions"])
            alpha = mask8 / 255.0                                          # float64, like the reference's numpy arithmetic
            img = alpha * img8 + (1.0 - alpha) * bg[None, None, :]
            if self.target_size is not None:                               # dataset/train.py:158-163
                img = iop.resize(img, self.target_size, interpolation=iop.INTER_LANCZOS4)
                alpha = iop.resize(alpha, self.target_size, interpolation=iop.INTER_LINEAR)
                scale_w, scale_h = self.target_size[0] / orig_W, self.target_size[1] / orig_H
            else:
                scale_w, scale_h = self.resize_img_scale
                if self.resize_img_scale != (1.0, 1.0):                    # dataset/train.py:165-173
                    img = iop.resize(img, None, fx=scale_w, fy=scale_h, interpolation=iop.INTER_LANCZOS4)
                    alpha = iop.resize(alpha, None, fx=scale_w, fy=scale_h, interpolation=iop.INTER_LINEAR)
            img = (img / 255.0).astype(np.float32)
            K[:1] *= scale_w                                               # dataset/train.py:239-244
            K[1:2] *= scale_h
            if self.crop_size != (-1, -1):                                 # dataset/train.py:255-256
                img, alpha, K = iop.crop_image(img, alpha, K, self.crop_size)
            out["K"] = K.astype(np.float32)
            out["target_rgbs"] = img
            out["target_masks"] = alpha[:, :, 0].astype(np.float32)
        return out


def write_synthetic_dataset(path: str, n_frames: int = 4, img: int = 64, level: int = 2) -> None:
    """A dataset directory in the reference's layout from the synthetic body (tests, demos)."""
    os.makedirs(os.path.join(path, "images"), exist_ok=True)
    os.makedirs(os.path.join(path, "masks"), exist_ok=True)
    body = _syn.icosphere_body(level)
    joints = _syn.TPOSE_JOINTS
    with open(os.path.join(path, "canonical_joints.pkl"), "wb") as f:
        pickle.dump({"joints": joints, "vertex": body["canonical_vertex"], "weights": body["canonical_lbs_weights"], "faces": body["faces"]}, f)
    cams, infos = {}, {}
    for i in range(n_frames):
        name = f"frame_{i:06d}"
        K, E = _syn.look_at_camera(img, yaw=2 * np.pi * i / max(n_frames, 1))
        cams[name] = {"intrinsics": K, "extrinsics": E}
        pose = _syn.random_pose(i)
        infos[name] = {"Rh": np.zeros(3, np.float32), "Th": np.zeros(3, np.float32), "poses": pose, "joints": joints, "tpose_joints": joints}
        try:
            from PIL import Image
            rng = np.random.default_rng(i)
            Image.fromarray((rng.uniform(0, 255, (img, img, 3))).astype(np.uint8)).save(os.path.join(path, "images", name + ".png"))
            m = np.zeros((img, img, 3), np.uint8); m[img // 4: 3 * img // 4, img // 3: 2 * img // 3] = 255
            Image.fromarray(m).save(os.path.join(path, "masks", name + ".png"))
        except ImportError:
            pass
    with open(os.path.join(path, "cameras.pkl"), "wb") as f:
        pickle.dump(cams, f)
    with open(os.path.join(path, "mesh_infos.pkl"), "wb") as f:
        pickle.dump(infos, f)
