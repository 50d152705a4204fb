"""On the GPU box: how many lanes of the segment kernels' evaluation loops do useful work on the metric workload, with the 8 x 8 quadrant
cull they use and with a 4 x 4 block cull (16-lane groups of a wave walking their own survivor lists) -- the measurement behind DESIGN.md
section 9.  Exact alphas of every (list entry, pixel of its tile) from the exported state; an entry `survives` a block when any pixel
centre in it reaches alpha >= 1/255 (the kernels' conservative test passes a few more)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd import _lib
from gomavatar_amd.workload import MetricWorkload

dev = "cuda"
wl = MetricWorkload(dev, subdiv=1, img=512, n_frames=8)
step = wl.step(1); bt = wl.batches(step)[0]
step.cam = bt["cam"]
step.forward_backward(wl.params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=False)
torch.cuda.synchronize()
P = step.F
H = W = 512; gx = gy = 32
st = step.state
D = int(st.poll()[0])
xy = st.export(_lib.BUF_XY, torch.empty(P, 2, device=dev))
co = st.export(_lib.BUF_CONIC_OPACITY, torch.empty(P, 4, device=dev))
tb = st.export(_lib.BUF_TILE_BASE, torch.empty(gx * gy + 1, dtype=torch.int32, device=dev)).long()
pl = st.export(_lib.BUF_POINT_LIST, torch.empty(D, dtype=torch.int32, device=dev)).long()
nc = st.export(_lib.BUF_N_CONTRIB, torch.empty(H, W, dtype=torch.int32, device=dev)).long()
cnt = tb[1:] - tb[:-1]
tile_of = torch.repeat_interleave(torch.arange(gx * gy, device=dev), cnt)
pos = torch.arange(D, device=dev) - tb[tile_of]
tile_nmax = nc.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gx * gy, 256).max(1).values
live = pos < tile_nmax[tile_of]                       # entries in front of the tile's last contributor (the passes stop behind it)
ent = torch.nonzero(live).squeeze(1)
print(f"frame 0: D = {D}, entries in front of the last contributor: {ent.numel()}")
py, px = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
blk4 = ((py // 4) * 4 + px // 4).reshape(-1)          # 4 x 4 block of a pixel (16 per tile)
q8 = ((py // 8) * 2 + px // 8).reshape(-1)            # 8 x 8 quadrant
blk_q = torch.tensor([q8[(blk4 == b).nonzero()[0, 0]].item() for b in range(16)], device=dev)
useful = lanes8 = lanes4 = 0
key_parts, s8_parts, s4_parts = [], [], []
for c0 in range(0, ent.numel(), 1 << 15):
    e = ent[c0:c0 + (1 << 15)]
    g = pl[e]; t = tile_of[e]
    fx = ((t % gx) * 16)[:, None] + px.reshape(1, -1).float()
    fy = ((t // gx) * 16)[:, None] + py.reshape(1, -1).float()
    dx = xy[g, 0:1] - fx; dy = xy[g, 1:2] - fy
    power = -0.5 * (co[g, 0:1] * dx * dx + co[g, 2:3] * dy * dy) - co[g, 1:2] * dx * dy
    alpha = torch.clamp(co[g, 3:4] * torch.exp(power), max=0.99)
    u = (power <= 0) & (alpha >= 1.0 / 255.0)                                   # [n, 256]
    s8 = torch.stack([u[:, q8 == q].any(1) for q in range(4)], 1)               # [n, 4]
    s4 = torch.stack([u[:, blk4 == b].any(1) for b in range(16)], 1)            # [n, 16]
    useful += int(u.sum()); lanes8 += int(s8.sum()) * 64; lanes4 += int(s4.sum()) * 16
    key_parts.append(t * 64 + pos[e] // 64); s8_parts.append(s8); s4_parts.append(s4)
key = torch.cat(key_parts); s8 = torch.cat(s8_parts).long(); s4 = torch.cat(s4_parts).long()
uk, inv = torch.unique(key, return_inverse=True)
it8 = torch.zeros(uk.numel(), 4, dtype=torch.long, device=dev).index_add_(0, inv, s8)       # survivors per (tile, sub-range, quadrant)
it4 = torch.zeros(uk.numel(), 16, dtype=torch.long, device=dev).index_add_(0, inv, s4)      # ... per 4 x 4 block
it4_wave = torch.stack([it4[:, blk_q == q].max(1).values for q in range(4)], 1)             # a wave = a quadrant: as long as its busiest group
print(f"lanes evaluated, 8x8 cull: {lanes8:.3e}, useful {useful / lanes8:.3f}")
print(f"lanes evaluated, 4x4 cull: {lanes4:.3e}, useful {useful / lanes4:.3f}  ({lanes8 / lanes4:.2f}x fewer)")
print(f"loop trips per (sub-range, quadrant) wave: 8x8 {it8.sum().item()}, 4x4 groups (busiest of the wave's four) {it4_wave.sum().item()} "
      f"= {it4_wave.sum().item() / it8.sum().item():.3f} of today's; mean of the four {it4.sum().item() / 4 / it8.sum().item():.3f}")
wg8 = it8.max(1).values.sum().item(); wg4 = it4_wave.max(1).values.sum().item()
print(f"per workgroup (four waves wait for the busiest): 8x8 {wg8}, 4x4 {wg4} = {wg4 / wg8:.3f}")
