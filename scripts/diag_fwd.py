"""Per-wave cycle breakdown of the render forward (instrumented build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gomavatar_amd import _lib, rasterizer as R, synthetic as syn
from gomavatar_amd.pipeline import RenderStep
subdiv = int(os.environ.get("SUBDIV", "1")); img = 512
body = syn.make_body(subdiv); N, F = body["canonical_vertex"].shape[0], body["faces"].shape[0]
w = torch.from_numpy(body["canonical_lbs_weights"]).T; w25 = torch.cat([w, torch.zeros(1, N)], 0).contiguous()
step = RenderStep(torch.from_numpy(body["faces"]), N, (img, img), w25)
gp = syn.make_gaussian_params(F, 1)
params = dict(vertices=torch.from_numpy(body["canonical_vertex"]).T.contiguous().cuda(), so3=torch.from_numpy(gp["so3"]).cuda(), scale=torch.from_numpy(gp["scale"]).cuda(), appearance=torch.from_numpy(gp["appearance"]).cuda())
fr = syn.make_frame(0, img)
d = {k: torch.from_numpy(fr[k][0]).contiguous().cuda() for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
step.set_camera(fr["K"][0], fr["E"][0])
gt = torch.zeros((img, img, 3), device="cuda"); gm = torch.zeros((img, img), device="cuda"); bg = torch.zeros(3, device="cuda")
step.state.set_option(_lib.OPT_PROFILE, 1)
for _ in range(3):
    step.forward_backward(params, d, gt, gm, bg)
torch.cuda.synchronize()
print("kernel ms", step.state.kernel_times_ms())
lib = _lib.load()
if hasattr(lib, "gom_debug_fetch"):
    n = 8192 * 8
    buf = (ctypes.c_ulonglong * n)()
    lib.gom_debug_fetch(buf, n)
    allb = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    a = allb[: 1024 * 4].reshape(1024, 4); ph = allb[8192 * 4: 8192 * 4 + 1024 * 3].reshape(1024, 3)
    order = np.argsort(-(a[:, 1]))[:8]
    print("tile n load sort write | dpp lds reg")
    for t in order:
        print(t, a[t, 3], a[t, 0], a[t, 1], a[t, 2], "|", ph[t, 0], ph[t, 1], ph[t, 2])
