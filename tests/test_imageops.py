"""Host-side image operations of the dataset reader (gomavatar_amd/imageops.py): numpy restatements of cv2.undistort / cv2.resize
as dataset/train.py:138-173 uses them.  OpenCV is absent: these are property and hand-computed checks (parity unpinned)."""
import numpy as np
import pytest

from gomavatar_amd import imageops as io


def _K(f=100.0, c=(31.5, 23.5)):
    return np.array([[f, 0, c[0]], [0, f, c[1]], [0, 0, 1]], dtype=np.float64)


def test_undistort_with_zero_coefficients_is_the_identity():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    out = io.undistort(img, _K(), np.zeros(5))
    assert out.dtype == np.uint8 and np.array_equal(out, img)
    assert np.array_equal(io.undistort(img[..., 0], _K(), np.zeros(5)), img[..., 0])


def test_undistort_map_matches_the_brown_conrady_model_by_hand():
    K, D = _K(f=50.0, c=(10.0, 8.0)), np.array([0.1, -0.02, 0.003, -0.004, 0.01])
    mx, my = io.undistort_map(K, D, 16, 20)
    u, v = 17, 3
    x, y = (u - 10.0) / 50.0, (v - 8.0) / 50.0
    r2 = x * x + y * y
    kr = 1 + 0.1 * r2 - 0.02 * r2 ** 2 + 0.01 * r2 ** 3
    xd = x * kr + 2 * 0.003 * x * y - 0.004 * (r2 + 2 * x * x)
    yd = y * kr + 0.003 * (r2 + 2 * y * y) + 2 * -0.004 * x * y
    assert abs(mx[v, u] - (50 * xd + 10)) < 1e-4 and abs(my[v, u] - (50 * yd + 8)) < 1e-4


def test_remap_is_bilinear_on_the_32nd_pixel_grid_with_zero_border():
    img = np.zeros((4, 4), np.uint8)
    img[1, 1], img[1, 2], img[2, 1], img[2, 2] = 100, 200, 40, 80
    mx = np.array([[1.25, 1.0, -0.5, 3.5]], np.float32)
    my = np.array([[1.5, 1.0, 1.0, 3.0]], np.float32)
    out = io.remap_linear_u8(img, mx, my)
    a = 0.25
    expect0 = 0.5 * ((1 - a) * 100 + a * 200) + 0.5 * ((1 - a) * 40 + a * 80)
    assert out[0, 0] == int(np.floor(expect0 + 0.5)) and out[0, 1] == 100
    assert out[0, 2] == 0 and out[0, 3] == 0            # half / fully outside: the outside taps are the constant border 0
    # a position between grid points is rounded to the nearest 1/32
    o2 = io.remap_linear_u8(img, np.array([[1.0 + 1 / 64 + 1e-3]], np.float32), np.array([[1.0]], np.float32))
    assert o2[0, 0] == int(np.floor(100 * 31 / 32 + 200 / 32 + 0.5))


def test_barrel_distortion_pulls_the_corners_inwards():
    img = np.zeros((64, 64), np.uint8); img[:, :] = 255
    out = io.undistort(img, _K(f=40.0, c=(31.5, 31.5)), np.array([0.4, 0.0, 0.0, 0.0, 0.0]))
    assert out[32, 32] == 255 and out[0, 0] == 0        # the corner samples outside the source image


@pytest.mark.parametrize("kind", [io.INTER_LINEAR, io.INTER_LANCZOS4])
def test_resize_keeps_constants_and_shapes(kind):
    img = np.full((30, 40, 3), 0.37, np.float64)
    out = io.resize(img, (16, 12), interpolation=kind)
    assert out.shape == (12, 16, 3) and out.dtype == np.float64 and np.allclose(out, 0.37, atol=1e-6)
    m = np.full((30, 40), 0.5, np.float32)
    assert io.resize(m, None, fx=0.5, fy=0.5, interpolation=kind).shape == (15, 20)


def test_lanczos_coefficients_sum_to_one_and_interpolate():
    c = io._lanczos4_coeffs(np.array([0.0, 0.25, 0.5, 0.999], np.float32))
    assert np.allclose(c.sum(-1), 1.0, atol=1e-6)
    assert np.array_equal(c[0], np.array([0, 0, 0, 1, 0, 0, 0, 0], np.float32))
    assert abs(c[2, 3] - c[2, 4]) < 1e-6 and c[2, 3] > 0.55       # symmetric around the midpoint, main lobe


def test_lanczos_resize_of_a_ramp_stays_a_ramp_inside():
    x = np.arange(64, dtype=np.float64)
    img = np.tile(x[None, :, None], (8, 1, 1))
    out = io.resize(img, (32, 8), interpolation=io.INTER_LANCZOS4)
    expect = (np.arange(32) + 0.5) * 2 - 0.5
    assert np.allclose(out[4, 4:-4, 0], expect[4:-4], atol=2e-3)   # a windowed sinc reproduces linear functions up to its ripple


def test_linear_resize_by_hand_and_the_two_by_two_area_path():
    img = np.arange(12, dtype=np.float32).reshape(1, 12)
    up = io.resize(np.tile(img, (2, 1)), (24, 2), interpolation=io.INTER_LINEAR)     # x2 upsampling: taps at (d + 0.5) / 2 - 0.5
    assert up.shape == (2, 24) and up[0, 0] == 0.0 and abs(up[0, 1] - 0.25) < 1e-6 and abs(up[0, 2] - 0.75) < 1e-6 and up[0, 23] == 11.0
    sq = np.arange(16, dtype=np.float64).reshape(4, 4)
    dn = io.resize(sq, (2, 2), interpolation=io.INTER_LINEAR)                        # exact 2 x 2 decimation = block means
    assert np.array_equal(dn, np.array([[2.5, 4.5], [10.5, 12.5]]))
    dn3 = io.resize(np.tile(np.arange(9, dtype=np.float64)[None], (3, 1)), (3, 3), interpolation=io.INTER_LINEAR)   # 3x: plain 2-tap at 1, 4, 7
    assert np.allclose(dn3[0], [1.0, 4.0, 7.0])


def test_crop_follows_the_reference_rules():
    rng = np.random.RandomState(0)
    img = np.zeros((120, 160, 3), np.float32); mask = np.zeros((120, 160, 1), np.float32)
    mask[40:80, 60:100] = 1.0
    K = _K(c=(80.0, 60.0))
    for _ in range(5):
        ci, cm, Kn = io.crop_image(img, mask, K, (64, 48), rng)
        assert ci.shape == (48, 64, 3) and cm.shape == (48, 64, 1) and cm.sum() >= 20
        ox, oy = K[0, 2] - Kn[0, 2], K[1, 2] - Kn[1, 2]
        assert 0 <= ox <= 160 - 64 and 0 <= oy <= 120 - 48 and np.array_equal(cm, mask[int(oy):int(oy) + 48, int(ox):int(ox) + 64])
