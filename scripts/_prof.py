import sys, time, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["x", "--iters", "0", "--level", "1", "--img", "512"]
exec(open("/root/repo/scripts/train_synthetic.py").read().split("log, t0 = [], time.perf_counter()")[0])
def it(i):
    fr = frames[i % 8]
    opt.zero_grad(set_to_none=True)
    rgbs, masks, out = student(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], i_iter=i)
    pred = unpack(rgbs, masks, fr["bgcolor"])
    total, losses = compute_loss(pred, masks, out, fr["gt_rgb"], fr["gt_mask"], loss_cfg, lpips_func=lp)
    total.backward(); opt.step()
for i in range(5): it(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(N): it(i)
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = [(e.key, e.count / N, e.device_time_total / N) for e in ev if e.device_time_total > 0 and e.self_device_time_total > 0 and not e.key.startswith("aten::") and not e.key.startswith("autograd") and "Backward" not in e.key]
rows.sort(key=lambda r: -r[1])
print("kernels per iteration: %.0f, GPU us per iteration: %.0f" % (sum(r[1] for r in rows), sum(r[2] for r in rows)))
for k, c, t in rows[:28]:
    print("%6.1f x %8.1f us  %s" % (c, t, k[:110]))
cpu = [(e.key, e.count / N, e.self_cpu_time_total / N) for e in ev if e.key.startswith("aten::")]
cpu.sort(key=lambda r: -r[2])
print("top aten ops by self CPU time per iteration:")
for k, c, t in cpu[:22]:
    print("%6.1f x %8.1f us  %s" % (c, t, k))
