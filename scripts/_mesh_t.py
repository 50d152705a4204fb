import sys, time, torch
sys.path.insert(0, "/root/repo")
from gomavatar_amd import synthetic as syn
from gomavatar_amd.mesh_renderer import MeshNormalRenderer, vertex_normals
from gomavatar_amd.geometry import MeshTopology
import numpy as np
body = syn.make_body(1); img = 512
fr = {k: torch.from_numpy(v).cuda() for k, v in syn.make_frame(0, img).items()}
v = torch.from_numpy(body["canonical_vertex"]).float().cuda()
faces = torch.from_numpy(body["faces"].astype(np.int64)).cuda()
r = MeshNormalRenderer((img, img), sigma=1e-5).train()
topo = r.topology(faces, v.shape[0])
def it():
    vv = v.clone().requires_grad_()
    vn = vertex_normals(vv, topo)
    n, a = r(vv.T[None], vn[None], fr["K"], fr["E"], faces)
    (n.sum() + a.sum()).backward()
for _ in range(5): it()
from gomavatar_amd import _lib
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): it()
torch.cuda.synchronize(); print("mesh branch fwd+bwd: %.1f us" % ((time.perf_counter() - t) / 50 * 1e6))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): it()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)[:6]:
    print("%8.1f us  %s" % (e.self_device_time_total / 10, e.key[:70]))
D, ov = r.state.poll()
print("mesh pairs D =", D, "-> segments of 128 >=", D // 128)
