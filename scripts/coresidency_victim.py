#!/usr/bin/env python
"""The STRONGEST victim (scripts/ubench/pkfma_victim.hip: nothing but exact `v_pk_fma_f32 ... op_sel:[0,1,0]` chains, with 16 KB of LDS or without any) beside the
PRODUCT's own kernels on a second stream of the same process: are the LPIPS bf16x3 trunk kernels (vgg_bf16.hip) aggressors in the sense of LABBOOK R6.8?  Positive
control: the opt-in matrix-core shadow MLP (mlp_mc.hip).  usage: python scripts/coresidency_victim.py /tmp/libpkvictim.so [seconds]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomavatar_amd.lpips import LPIPSMatrixCore
from gomavatar_amd.model import ShadowModule, _ShadeUnderMesh

vlib = ctypes.CDLL(sys.argv[1])
vlib.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = "cuda"
side = torch.cuda.Stream()
sm = ShadowModule().to(dev)
sm_params = [p for m in sm.block_mlps if isinstance(m, torch.nn.Linear) for p in (m.weight, m.bias)]

def aggressor(kind, img):
    if kind == "none":
        return lambda: None
    if kind in ("mc", "valu_mlp"):
        nrm = torch.nn.functional.normalize(torch.randn(img * img, 3, device=dev), dim=-1)
        def f():
            _ShadeUnderMesh.matrix_cores = kind == "mc"
            x = nrm.clone().requires_grad_()
            _ShadeUnderMesh.apply(x, sm.multires, *sm_params).sum().backward()
            _ShadeUnderMesh.matrix_cores = False
        return f
    lp = LPIPSMatrixCore(trunk_seed=0, device=dev, precision=kind)
    a, b = torch.rand(1, img, img, 3, device=dev), torch.rand(1, img, img, 3, device=dev)
    def f():
        x = a.clone().requires_grad_()
        lp.loss(x, b).sum().backward()
    return f

for kind, img in (("none", 0), ("bf16x3", 512), ("bf16x3", 256), ("bf16", 512), ("valu_mlp", 256), ("mc", 256)):
    f = aggressor(kind, img)
    for with_lds in (1, 0):
        hist = torch.zeros(24, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.time(); launches = 0; evals = 0
        while time.time() - t0 < secs:
            with torch.cuda.stream(side):
                for _ in range(3 if kind not in ("mc", "valu_mlp") else 30):
                    f(); evals += 1
            for _ in range(40):
                vlib.pk_victim_launch(with_lds, 1024, 512, hist.data_ptr(), torch.cuda.current_stream().cuda_stream); launches += 1
            torch.cuda.synchronize()
        h = hist.cpu().reshape(6, 4)
        name = {"none": "no aggressor", "mc": "matrix-core shadow MLP (mlp_mc.hip, opt-in)", "valu_mlp": "fp32 VALU shadow MLP (mlp.hip, the default), forward + backward"}.get(kind, f"LPIPS trunk {kind}, {img}^2 (vgg_bf16.hip), forward + backward")
        print(f"{name:75s} victim {'with 16 KB LDS' if with_lds else 'without LDS   '}: {launches} launches beside {evals} aggressor calls; wrong results "
              f"(op_sel:[0,1,0] low half, lanes 48-63): {int(h[0, 3])}; anywhere else: {int(h.sum() - h[0, 3])}", flush=True)
