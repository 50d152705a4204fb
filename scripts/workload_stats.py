"""Workload statistics of the metric frame from the CPU oracle (test infrastructure; not shipped):
list lengths, per-pixel contributor counts, how many (entry, quadrant) evaluations are useful.
Usage: python scripts/workload_stats.py [subdiv] [img] [frame]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import body_scene
from oracle import geometry as og, raster as orast

sub = int(sys.argv[1]) if len(sys.argv) > 1 else 1
img = int(sys.argv[2]) if len(sys.argv) > 2 else 512
frame = int(sys.argv[3]) if len(sys.argv) > 3 else 0
orast.set_threads(8)
sc = body_scene(sub, frame, img)
with torch.no_grad():
    rgb, mask, aux = og.render_path(sc["params"], sc["frame"], sc["faces"], sc["lbs_weights"], img)
F = sc["faces"].shape[0]
feat = torch.cat([sc["params"]["appearance"].T, torch.ones(F, 1)], -1).numpy()
f = orast.forward(aux["cam"], aux["xyz"].numpy(), aux["cov6"].numpy(), feat, np.ones(F, np.float32))
D = f["D"]; rng = f["ranges"]; cnt = (rng[:, 1] - rng[:, 0]).astype(np.int64)
print("P", F, "D", D, "nonempty tiles", int((cnt > 0).sum()), "max list", int(cnt.max()), "mean nonempty", float(cnt[cnt > 0].mean()))
print("list length percentiles (nonempty):", np.percentile(cnt[cnt > 0], [10, 25, 50, 75, 90, 99]).astype(int))
print("tiles_touched mean (visible)", float(f["tiles_touched"][f["radii"] > 0].mean()), "visible", int((f["radii"] > 0).sum()))
nc = f["n_contrib"]
gx = (img + 15) // 16
xy = f["xy"]; co = f["conic_opacity"]; pl = f["point_list"]
tot_eval_tile = 0; tot_contrib = 0; tot_alpha_pairs = 0
q_surv = 0; q_surv_below = 0; q_useful = 0; q_total = 0
ent_with_contrib = 0; live_entries = 0
contrib_per_pixel = []
for t in np.nonzero(cnt > 0)[0]:
    tx, ty = t % gx, t // gx
    s, e = int(rng[t, 0]), int(rng[t, 1])
    g = pl[s:e]
    ex, ey = xy[g, 0], xy[g, 1]; a, b, c, o = co[g, 0], co[g, 1], co[g, 2], co[g, 3]
    px = (tx * 16 + np.arange(16))[None, :].repeat(16, 0).reshape(-1).astype(np.float32)
    py = (ty * 16 + np.arange(16))[:, None].repeat(16, 1).reshape(-1).astype(np.float32)
    dx = ex[:, None] - px[None, :]; dy = ey[:, None] - py[None, :]
    power = -0.5 * (a[:, None] * dx * dx + c[:, None] * dy * dy) - b[:, None] * dx * dy
    al = np.minimum(0.99, o[:, None] * np.exp(power)); al[(power > 0) | (al < 1 / 255)] = 0
    ncp = nc[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16].reshape(-1).astype(np.int64)
    idx = np.arange(e - s)[:, None]
    live = (idx < ncp[None, :])           # entry still ahead of the pixel's last contributor
    contrib = (al > 0) & live
    tot_alpha_pairs += int((al > 0).sum()); tot_contrib += int(contrib.sum())
    ent_with_contrib += int(contrib.any(1).sum()); live_entries += int((idx[:, 0] < ncp.max()).sum())
    contrib_per_pixel.append(contrib.sum(0))
    quad = ((np.arange(256) // 16) // 8) * 2 + ((np.arange(256) % 16) // 8)
    for q in range(4):
        m = quad == q
        surv = (al[:, m] > 0).any(1)            # what an exact quadrant cull keeps
        wmax = ncp[m].max()
        below = surv & (idx[:, 0] < wmax)
        useful = contrib[:, m].any(1)
        q_total += e - s; q_surv += int(surv.sum()); q_surv_below += int(below.sum()); q_useful += int(useful.sum())
cp = np.concatenate(contrib_per_pixel)
print("pixels with contributors", int((cp > 0).sum()), "mean contributors/pixel (covered)", float(cp[cp > 0].mean()), "p99", int(np.percentile(cp[cp > 0], 99)))
print("pairs alpha>0 (no termination)", tot_alpha_pairs, " contributing pairs (bwd useful)", tot_contrib)
print("tile entries: total", D, "below tile nmax", live_entries, "with >=1 contributing pixel", ent_with_contrib)
print("(entry,quadrant): total", q_total, "exact-cull survivors", q_surv, "survivors below wmax", q_surv_below, "with a contributing pixel", q_useful)
print("lane utilisation of useful (entry,quadrant): contributing pairs / (64*useful) =", tot_contrib / (64.0 * q_useful))
