"""gomavatar_amd.camera.camera_block -- the ONE camera function behind pipeline.RenderStep, model.Model and
rasterizer.DeviceCamera -- against the golden recorded through a stub of the reference's own Renderer.forward
(models/modules/renderer/gaussian.py:28-66; tests/golden/renderer_camera.npz): bit-exact on the host and on the device."""
import os

import numpy as np
import pytest
import torch

from gomavatar_amd.camera import camera_block


def _check(g, tanfov, view, proj, campos=None):
    assert np.float32(g["tanfovx"]) == np.float32(tanfov[0].item()) and np.float32(g["tanfovy"]) == np.float32(tanfov[1].item())
    np.testing.assert_array_equal(view.cpu().numpy(), g["viewmatrix"])
    np.testing.assert_array_equal(proj.cpu().numpy(), g["projmatrix"])
    if campos is not None:
        np.testing.assert_allclose(campos.cpu().numpy(), g["campos"], atol=1e-6)


def test_camera_block_host_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "renderer_camera.npz"))
    _check(g, *camera_block(torch.from_numpy(g["K"]), torch.from_numpy(g["E"]), 512, 512, want_campos=True))


def test_camera_block_nonsquare_and_offcentre_matches_float64_formula():
    rng = np.random.default_rng(0)
    for (H, W) in ((384, 512), (540, 540), (1024, 1024)):
        K = np.array([[1100.0 + rng.normal(), 0, W / 2 + 3.3], [0, 1230.0, H / 2 - 7.1], [0, 0, 1]], np.float32)
        E = np.eye(4, dtype=np.float32); E[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32); E[:3, 3] = rng.normal(size=3)
        tanfov, view, proj = camera_block(torch.from_numpy(K), torch.from_numpy(E), H, W)
        fx, fy, px, py = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
        zf, zn = 100, 0.001
        Kn = torch.tensor([[2 * fx / W, 0, (2 * px - W) / W, 0], [0, 2 * fy / H, (2 * py - H) / H, 0], [0, 0, zf / (zf - zn), -zf * zn / (zf - zn)],
                           [0, 0, 1, 0]]).float()      # gaussian.py:41-46: python floats -> float32 tensor
        ref = torch.from_numpy(E).T @ Kn.T             # gaussian.py:61
        assert float((proj - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        assert abs(float(tanfov[0]) - W / (2 * fx)) <= 1e-7 and abs(float(tanfov[1]) - H / (2 * fy)) <= 1e-7


@pytest.mark.gpu
def test_camera_block_device_and_every_consumer_match_reference_golden(golden_dir):
    import ctypes
    from gomavatar_amd import _lib, synthetic as syn
    from gomavatar_amd.pipeline import RenderStep
    from gomavatar_amd.rasterizer import DeviceCamera
    g = np.load(os.path.join(golden_dir, "renderer_camera.npz"))
    K, E = torch.from_numpy(g["K"]), torch.from_numpy(g["E"])
    # the function itself on the device
    _check(g, *camera_block(K.cuda(), E.cuda(), 512, 512, want_campos=True))

    def check_struct(cam):
        assert (cam.H, cam.W) == (512, 512)
        assert np.float32(cam.tanfovx) == np.float32(g["tanfovx"]) and np.float32(cam.tanfovy) == np.float32(g["tanfovy"])
        np.testing.assert_array_equal(np.array(cam.view[:], np.float32).reshape(4, 4), g["viewmatrix"])
        np.testing.assert_array_equal(np.array(cam.proj[:], np.float32).reshape(4, 4), g["projmatrix"])

    # consumer 1: pipeline.RenderStep (host struct + the device camera array of a batched step)
    body = syn.icosphere_body(1)
    N = body["canonical_vertex"].shape[0]
    w = torch.from_numpy(body["canonical_lbs_weights"]).T
    w25 = torch.cat([w, torch.zeros(1, N)], 0).contiguous()
    for B in (1, 2):
        step = RenderStep(torch.from_numpy(body["faces"]), N, (512, 512), w25, batch=B)
        step.set_camera(g["K"], g["E"])
        check_struct(step.cam)
        if B > 1:
            torch.cuda.synchronize()
            raw = step.cams_dev.cpu().numpy()
            for b in range(B):
                check_struct(_lib.GomCamera.from_buffer_copy(raw[b].tobytes()))
    # consumer 2: model.Model._camera
    from gomavatar_amd.model import Model
    from types import SimpleNamespace as NS
    m = Model(NS(img_size=(512, 512)), dict(faces=body["faces"], canonical_vertex=body["canonical_vertex"], canonical_lbs_weights=body["canonical_lbs_weights"]))
    check_struct(m._camera(K[None].cuda(), E[None].cuda(), (0.0, 0.0, 0.0, 0.0)))
    # consumer 3: rasterizer.DeviceCamera (160 bytes in the struct's layout, written by device ops only)
    dc = DeviceCamera(512, 512, "cuda").update(K.cuda(), E.cuda())
    raw = dc.data.cpu().numpy()
    assert raw[:2].view(np.int32).tolist() == [512, 512]
    assert raw[2] == np.float32(g["tanfovx"]) and raw[3] == np.float32(g["tanfovy"])
    np.testing.assert_array_equal(raw[4:20].reshape(4, 4), g["viewmatrix"])
    np.testing.assert_array_equal(raw[20:36].reshape(4, 4), g["projmatrix"])
    # consumer 4: a camera of ONE forward's own, every byte written by the native launch (gom_camera_update_device: what Model.forward uses for fp32 device K / E)
    bg4 = torch.tensor([0.1, 0.2, 0.3, 0.0], device="cuda")
    fc = DeviceCamera.fresh(512, 512, K.cuda(), E.cuda(), bg4)
    raw = fc.data.cpu().numpy()
    check_struct(_lib.GomCamera.from_buffer_copy(raw.tobytes()))
    assert raw[36:40].tolist() == bg4.cpu().numpy().tolist()
    # ... and over random cameras against the host function (fp64 tan(atan(.)) rounded to fp32, products summed without contraction): the same bits
    rng = np.random.default_rng(5)
    for _ in range(200):
        Kr = np.array([[rng.uniform(200, 3000), 0, rng.uniform(100, 900)], [0, rng.uniform(200, 3000), rng.uniform(100, 900)], [0, 0, 1]], np.float32)
        Er = np.eye(4, dtype=np.float32)
        Er[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        Er[:3, 3] = rng.normal(size=3).astype(np.float32) * 3
        Hh, Ww = int(rng.integers(64, 1100)), int(rng.integers(64, 1100))
        tf, vw, pj = camera_block(torch.from_numpy(Kr), torch.from_numpy(Er), Hh, Ww)
        raw = DeviceCamera.fresh(Hh, Ww, torch.from_numpy(Kr).cuda(), torch.from_numpy(Er).cuda(), bg4).data.cpu().numpy()
        assert raw[:2].view(np.int32).tolist() == [Hh, Ww]
        np.testing.assert_array_equal(raw[2:4], tf.numpy())
        np.testing.assert_array_equal(raw[4:20].reshape(4, 4), vw.numpy())
        np.testing.assert_array_equal(raw[20:36].reshape(4, 4), pj.numpy())
