import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from gomavatar_amd import geometry as G
from helpers import body_scene
sc = body_scene(0, frame=0, img=128); p, fr = sc["params"], sc["frame"]
N = p["vertices"].shape[1]; F = sc["faces"].shape[0]
topo = G.MeshTopology(sc["faces"], N, device="cuda")
cg = torch.randn(F*3, 3, device="cuda")
out = G._csr_gather(cg, topo, N)
ref = torch.zeros(N, 3, device="cuda").index_add_(0, sc["faces"].reshape(-1).cuda(), cg).T
print("csr_gather err", float((out-ref).abs().max()), float(ref.abs().max()))
# LBS backward alone
v = p["vertices"].cuda().requires_grad_()
R, T = G.get_global_RTs(fr["cnl_gtfms"].cuda(), fr["dst_Rs"].cuda(), fr["dst_Ts"].cuda())
vo = G.apply_lbs(v[None], R, T, sc["lbs_weights"].cuda())[0]
g = torch.randn_like(vo)
(vo*g).sum().backward()
from oracle import geometry as og
v2 = p["vertices"].clone().requires_grad_()
R2, T2 = og.fk_global_RTs(fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
vo2 = og.lbs(v2[None], R2, T2, sc["lbs_weights"])[0]
(vo2*g.cpu()).sum().backward()
print("lbs fwd err", float((vo.detach().cpu()-vo2.detach()).abs().max()), "bwd err", float((v.grad.cpu()-v2.grad).abs().max()), float(v2.grad.abs().max()))
# with R requiring grad
dR = fr["dst_Rs"].cuda().requires_grad_(); v3 = p["vertices"].cuda().requires_grad_()
R, T = G.get_global_RTs(fr["cnl_gtfms"].cuda(), dR, fr["dst_Ts"].cuda())
vo = G.apply_lbs(v3[None], R, T, sc["lbs_weights"].cuda())[0]
(vo*g).sum().backward()
print("lbs bwd (pose grad on) err", float((v3.grad.cpu()-v2.grad).abs().max()))
# fused vs unfused
def cu(t): return t.detach().clone().cuda()
v = cu(p["vertices"]).requires_grad_(); so3 = cu(p["so3"]).requires_grad_(); scl = cu(p["scale"]).requires_grad_()
R, T = G.get_global_RTs(fr["cnl_gtfms"].cuda(), fr["dst_Rs"].cuda(), fr["dst_Ts"].cuda())
vo = G.apply_lbs(v[None], R, T, sc["lbs_weights"].cuda())[0]
vo.retain_grad()
xyz, cov6 = G.face_gaussians(vo, so3, scl, topo)
(xyz.sum() + cov6.sum() * 100).backward()
v2 = cu(p["vertices"]).requires_grad_(); so32 = cu(p["so3"]).requires_grad_(); scl2 = cu(p["scale"]).requires_grad_()
x2, c2, vo2 = G.posed_face_gaussians(v2, so32, scl2, fr["dst_Rs"].cuda(), fr["dst_Ts"].cuda(), fr["cnl_gtfms"].cuda(), sc["lbs_weights"].cuda(), topo)
(x2.sum() + c2.sum() * 100).backward()
print("fwd equal", torch.equal(xyz, x2), torch.equal(cov6, c2), torch.equal(vo, vo2))
print("so3 grad diff", float((so3.grad-so32.grad).abs().max()), "scale", float((scl.grad-scl2.grad).abs().max()))
print("v grad diff", float((v.grad-v2.grad).abs().max()), float(v2.grad.abs().max()), float(v.grad.abs().max()), "vo.grad max", float(vo.grad.abs().max()))
# oracle for this loss
po = [p["vertices"].double().requires_grad_(), p["so3"].double().requires_grad_(), p["scale"].double().requires_grad_()]
Rs, Ts = og.fk_global_RTs(fr["cnl_gtfms"].double(), fr["dst_Rs"].double(), fr["dst_Ts"].double())
vob = og.lbs(po[0][None], Rs, Ts, sc["lbs_weights"].double())[0]; vob.retain_grad()
xo, co = og.face_gaussians(vob, sc["faces"], po[1], po[2], 1e-3)
(xo.sum() + og.pack_cov6(co).sum()*100).backward()
print("oracle v grad max", float(po[0].grad.abs().max()), "vobs grad max", float(vob.grad.abs().max()))
print("unfused vs oracle", float((v.grad.cpu().double()-po[0].grad).abs().max()), "fused vs oracle", float((v2.grad.cpu().double()-po[0].grad).abs().max()))
print("unfused vo.grad vs oracle", float((vo.grad.cpu().double()-vob.grad).abs().max()))
