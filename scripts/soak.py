#!/usr/bin/env python
"""Randomised soak of the GPU parity tests: for a time budget, run the raster fuzz test (tests/test_gpu_raster.py) over new seeds and,
interleaved, random configurations of batch-equals-single-frames, the 3x3 convolution against torch and the fused shadow MLP against
the CPU module.  Prints every failure and a count.   python scripts/soak.py [seconds=300] [first_seed=10]

Round-1 record (MI355X, ~50 GPU-minutes): 12 000 raster scenes (every integer output bit-exact in all of them; 10 of the first
3 600 exceeded the image criteria by ONE threshold-flip pixel on images of a few thousand pixels, which the criteria now allow),
2 000 batch configurations, 1 300 convolution shapes, 1 700 MLP configurations, 250 mesh-raster scenes (5 with ONE pixel on the
rim of a blur band: now a flip count; 2 more while the pixel centres were computed with v_rcp: reverted to IEEE
division), 700 each of the L1-term / compose / unpack / NDC kernels (one tolerance that was an artefact of a 135 x 9 image).
Last runs on the round's final code: 14 + 22 minutes, 3 241 + 4 986 checks, no failure."""
import sys, os, time, traceback
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir("/root/repo")
import numpy as np, torch
import test_gpu_raster as TR, test_gpu_batch as TB, test_gpu_vgg_bf16 as TV, test_gpu_model as TM, test_gpu_mesh as TMe, test_gpu_metrics as TMt, test_gpu_loss as TL
t0 = time.time(); n_ok = n_bad = 0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(5)
while time.time() - t0 < budget:
    try:
        TR.test_fuzz_shapes_scales_and_depths(seed)
        n_ok += 1
    except Exception as e:
        n_bad += 1; print("FUZZ FAIL seed", seed, repr(e)[:300], flush=True)
    if seed % 6 == 0:
        B, img, graph, smpl = int(rng.integers(2, 9)), int(rng.choice([64, 96, 128, 160, 256])), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if smpl: img = 256
        try:
            TB.test_batch_equals_single_frames_bitwise(B, img, graph, smpl)
            n_ok += 1
        except AssertionError as e:
            # the smpl_like branch asserts a property of the 256x256 body scene only; other failures are real
            n_bad += 1; print("BATCH FAIL", B, img, graph, smpl, repr(e)[:300], flush=True); traceback.print_exc()
        except Exception as e:
            n_bad += 1; print("BATCH ERR", B, img, graph, smpl, repr(e)[:300], flush=True)
    if seed % 10 == 0:
        cfgc = (int(rng.integers(1, 4)), int(rng.integers(3, 300)), int(rng.integers(3, 300)), int(rng.choice([32, 64, 128, 256])), int(rng.choice([64, 128, 256])), bool(rng.integers(0, 2)))
        try:
            TV.test_conv3x3_matches_torch(*cfgc); n_ok += 1
        except Exception as e:
            n_bad += 1; print("CONV FAIL", cfgc, repr(e)[:300], flush=True)
        cfgm = (int(rng.choice([2, 4, 6])), int(rng.choice([32, 64, 96, 128])), 3, (4,), int(rng.integers(1, 9000)))
        try:
            TM.test_shadow_mlp_fused_kernels_vs_torch_cpu(*cfgm); n_ok += 1
        except Exception as e:
            n_bad += 1; print("MLP FAIL", cfgm, repr(e)[:300], flush=True)
    if seed % 5 == 0:
        H, W = int(rng.integers(9, 200)), int(rng.integers(9, 200))
        for name, fn, cfgl in (("L1TERMS", TL.test_l1_terms_match_torch_formulas, (H, W, int(rng.choice([1, 3, 5, 7, 9])), bool(rng.integers(0, 2)))),
                               ("COMPOSE", TL.test_compose_and_unpack_kernels_match_the_torch_chains, (H, W, bool(rng.integers(0, 2)))),
                               ("NDC", TMe.test_ndc_from_world_kernel_matches_the_torch_formula, (H, W))):
            try:
                fn(*cfgl); n_ok += 1
            except Exception as e:
                n_bad += 1; print(name, "FAIL", cfgl, repr(e)[:300], flush=True)
    if seed % 15 == 0:
        cfgs = (int(rng.integers(40, 120)), float(rng.uniform(0.8, 4.0)))
        try:
            TMe.test_forward_and_backward_match_oracle(*cfgs); n_ok += 1
        except Exception as e:
            n_bad += 1; print("MESH FAIL", cfgs, repr(e)[:300], flush=True)
        try:
            TMt.test_ssim_both_definitions_and_psnr(seed); n_ok += 1
        except Exception as e:
            n_bad += 1; print("SSIM FAIL", seed, repr(e)[:300], flush=True)
    seed += 1
print("soak: ok", n_ok, "bad", n_bad, "last seed", seed, "in", round(time.time() - t0), "s")
