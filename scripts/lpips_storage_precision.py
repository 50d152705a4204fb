"""CPU, float64: what rounding the post-ReLU ACTIVATIONS of the LPIPS-VGG trunk to 24 / 17 / 16 mantissa bits (everything else exact) does to
the LPIPS value and to its image gradient -- the experiment behind the bf16x3 trunk's gradient tolerance (tests/test_gpu_vgg_bf16.py):
the value moves by ~2e-7, the gradient by ~1e-3 .. 3e-3 relative L2 (max 2e-2 of the largest element), because thirteen ReLU masks and four
max-pool argmaxes branch on the activations.  fp32 storage (24 bits): 4e-7.   python scripts/lpips_storage_precision.py"""
import torch, torch.nn.functional as F, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd.lpips import seeded_trunk, POOL_BEFORE_CONV, TAP_AFTER_CONV, SHIFT, SCALE, _DATA
torch.set_num_threads(8)
B,H,W=2,64,96
g = torch.Generator().manual_seed(11)
pred = torch.rand(B, H, W, 3, generator=g)
gt = (pred + 0.2 * torch.randn(B, H, W, 3, generator=g)).clamp(0, 1)
wb=[t.double() for t in seeded_trunk(5)]
lin=np.load(_DATA); lins=[torch.from_numpy(lin[f"lin{k}"].astype(np.float64).reshape(-1)) for k in range(5)]
shift=torch.tensor(SHIFT,dtype=torch.float64).view(1,3,1,1); scale=torch.tensor(SCALE,dtype=torch.float64).view(1,3,1,1)
class Q(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bits):
        if bits is None: return x
        m, e = torch.frexp(x)
        return torch.ldexp(torch.round(m * 2**bits) / 2**bits, e)
    @staticmethod
    def backward(ctx, g): return g, None
def feats(x, bits):
    taps=[]; h=Q.apply((x-shift)/scale, bits)
    for i in range(13):
        if i in POOL_BEFORE_CONV: h=F.max_pool2d(h,2,2)
        h=Q.apply(F.relu(F.conv2d(h, wb[2*i], wb[2*i+1], padding=1)), bits)
        if i in TAP_AFTER_CONV: taps.append(h)
    return taps
def run(bits):
    p=pred.double().clone().requires_grad_()
    f0=feats(2*p.permute(0,3,1,2)-1, bits)
    with torch.no_grad(): f1=feats(2*gt.double().permute(0,3,1,2)-1, bits)
    val=0
    for k in range(5):
        n0=f0[k]/(f0[k].pow(2).sum(1,keepdim=True).sqrt()+1e-10); n1=f1[k]/(f1[k].pow(2).sum(1,keepdim=True).sqrt()+1e-10)
        val=val+((n0-n1)**2*lins[k].view(1,-1,1,1)).sum(1).mean((1,2))
    val=val.mean(); val.backward()
    return float(val), p.grad
v0,g0=run(None)
for bits in (24, 17, 16):
    v,gq=run(bits)
    e=(gq-g0).abs().flatten()
    print(bits, 'value rel', abs(v-v0)/v0, 'grad rel L2', float(e.norm()/g0.norm()), 'q99.9/max', float(torch.quantile(e,0.999)/g0.abs().max()), 'max/max', float(e.max()/g0.abs().max()))
