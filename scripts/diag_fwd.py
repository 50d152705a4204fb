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
    step.forward_backward(params, d, gt, gm, bg, backward=False); torch.cuda.synchronize()
    n = 8192 * 4
    buf = (ctypes.c_ulonglong * n)()
    lib.gom_debug_fetch(buf, n)
    nseg = min(8192, int(step.state.export(_lib.BUF_STATUS, torch.empty(4, dtype=torch.int32, device="cuda")).cpu().numpy()[2]))
    a = np.frombuffer(buf, dtype=np.uint64)[: nseg * 4].reshape(nseg, 4)
    t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64); base = t0.min()
    t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # us (100 MHz)
    hw = a[:, 2]; xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf; hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    dur = t1 - t0
    print("k_seg_T blocks", nseg, "span us", t1.max(), "block dur us mean/p50/p90/max", dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), dur.max())
    print("distinct CUs used", len(np.unique(cuid)), "blocks per CU max", np.bincount(np.unique(cuid, return_inverse=True)[1]).max())
    for T in (1, 3, 5, 8, 12, 16, 20, 25, 30):
        print("  t=%2d us: blocks running %d" % (T, int(((t0 <= T) & (t1 > T)).sum())))
    late = np.argsort(-t1)[:5]
    print("last finishers (start,end,dur,blockIdx):", [(round(t0[i], 1), round(t1[i], 1), round(dur[i], 1), int(a[i, 3])) for i in late])
    print("start time pct", np.percentile(t0, [1, 25, 50, 75, 99]))
