"""gomavatar_amd.optim.GomAdam -- torch.optim.Adam(Model.get_param_groups()) as ONE native launch (gom_adam_multi) -- against torch's own Adam:
same parameters and moments over steps with the reference's per-iteration learning-rate rewrite (update_lr, train.py:166-175), parameters
without a gradient skipped, checkpoints exchanged with torch.optim.Adam in both directions (state_dict layout), and the capturable variant
(device step counter) inside a replayed graph."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(3, 27554), (3, 55104), (3, 55104), (3, 55104), (128, 39), (128,), (128, 128), (128,), (1, 128), (1,), (25, 100)]
    ps = [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]
    return ps, [dict(name="lbs_weights", params=[ps[10]], lr=0.0), dict(name="appearance", params=[ps[3]], lr=5e-4), dict(name="xyz", params=[ps[0]], lr=5e-4),
                dict(name="geo", params=[ps[1]], lr=3e-4), dict(name="geo", params=[ps[2]], lr=3e-4), dict(name="shadow", params=ps[4:10], lr=1e-3)]


def _set_grads(ps, it):
    g = torch.Generator().manual_seed(100 + it)
    for k, p in enumerate(ps):
        p.grad = None if k == 10 else (torch.randn(p.shape, generator=g) * 10.0 ** ((k + it) % 5 - 3)).cuda()   # the buffer group never has a gradient


def test_gom_adam_follows_torch_adam_and_exchanges_checkpoints():
    from gomavatar_amd.optim import GomAdam
    pa, ga = _groups(0)
    pb, gb = _groups(0)
    oa = torch.optim.Adam(ga, betas=(0.9, 0.999))
    ob = GomAdam(gb, betas=(0.9, 0.999))
    assert [g["name"] for g in ob.param_groups] == [g["name"] for g in oa.param_groups]

    def close():
        for a, b in zip(pa, pb):
            assert float((a - b).detach().abs().max()) <= 2e-6 * float(a.detach().abs().max())
        for a, b in zip(pa, pb):
            if a in oa.state:
                for k in ("exp_avg", "exp_avg_sq"):
                    ra, rb = oa.state[a][k], ob.state[b][k]
                    assert float((ra - rb).abs().max()) <= 2e-6 * float(ra.abs().max()) + 1e-30, k
                assert float(oa.state[a]["step"]) == float(ob.state[b]["step"])

    for it in range(6):
        for ps, o in ((pa, oa), (pb, ob)):
            _set_grads(ps, it)
            o.step()
            for g in o.param_groups:                          # update_lr
                g["lr"] = {"lbs_weights": 0.0, "appearance": 5e-4, "xyz": 5e-4, "geo": 3e-4, "shadow": 1e-3}[g["name"]] * 0.1 ** (it / 7.0)
        close()
    assert pb[10] not in ob.state or len(ob.state[pb[10]]) == 0     # no gradient, no state (as torch)
    # checkpoints both ways: torch's state into GomAdam, GomAdam's state into torch -- then three more steps together
    import copy
    sd_a, sd_b = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())   # (as torch.save / torch.load would: load_state_dict itself does not copy tensors)
    assert sd_a["state"].keys() == sd_b["state"].keys() and all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in sd_b["state"].values())
    pc, gc = _groups(0)
    pd, gd = _groups(0)
    with torch.no_grad():
        for c, d, a in zip(pc, pd, pa):
            c.copy_(a); d.copy_(a)
    oc, od = GomAdam(gc, betas=(0.9, 0.999)), torch.optim.Adam(gd, betas=(0.9, 0.999))
    oc.load_state_dict(sd_a); od.load_state_dict(sd_b)
    for it in range(6, 9):
        for ps, o in ((pa, oa), (pc, oc), (pd, od)):
            _set_grads(ps, it)
            o.step()
    for a, c, d in zip(pa, pc, pd):
        assert float((a - c).detach().abs().max()) <= 2e-6 * float(a.detach().abs().max()) and float((a - d).detach().abs().max()) <= 2e-6 * float(a.detach().abs().max())


def test_gom_adam_capturable_in_a_replayed_graph():
    from gomavatar_amd.optim import GomAdam
    pa, ga = _groups(1)
    pb, gb = _groups(1)
    oa, ob = GomAdam(ga), GomAdam(gb, capturable=True)
    _set_grads(pa, 0); _set_grads(pb, 0)
    static = [p.grad for p in pb if p.grad is not None]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ob.step()                                              # warm-up step 1 (outside the graph)
        gr = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            ob.step()
    oa.step()
    # (the capture itself does not execute: replay = step 2, 3, 4 with fresh gradients written into the static buffers)
    for it in range(1, 4):
        _set_grads(pa, it)
        oa.step()
        g = torch.Generator().manual_seed(100 + it)
        k2 = 0
        for k, p in enumerate(pb):
            if k == 10:
                continue
            static[k2].copy_((torch.randn(p.shape, generator=g) * 10.0 ** ((k + it) % 5 - 3)).cuda()); k2 += 1
        gr.replay()
    torch.cuda.synchronize()
    assert int(ob._step_dev.item()) == 4
    for a, b in zip(pa, pb):
        assert float((a - b).detach().abs().max()) <= 2e-6 * float(a.detach().abs().max())
    # a checkpoint written after REPLAYS carries the device count (4), not the capture-time Python count (2) -- round-4 advisor finding -- and a fresh
    # optimizer loaded from it continues with step 5's bias corrections, like the eager optimizer that has really stepped four times
    sd = ob.state_dict()
    assert {int(float(st["step"])) for st in sd["state"].values()} == {4}
    pc, gc = _groups(1)
    with torch.no_grad():
        for c, b in zip(pc, pb):
            c.copy_(b)
    oc = GomAdam(gc, capturable=True)
    oc.load_state_dict(sd)
    _set_grads(pa, 4); _set_grads(pc, 4)
    oa.step(); oc.step()
    assert int(oc._step_dev.item()) == 5
    for a, c in zip(pa, pc):
        assert float((a - c).detach().abs().max()) <= 2e-6 * float(a.detach().abs().max())
    # a torch CAPTURABLE checkpoint holds `step` on the device: loaded onto the host, no synchronising read inside step()
    sd2 = oa.state_dict()
    for st in sd2["state"].values():
        st["step"] = st["step"].cuda()
    od = GomAdam(_groups(1)[1], capturable=True)
    od.load_state_dict(sd2)
    assert all(not st["step"].is_cuda for st in od.state.values())
    # parameters with different step counts cannot share the one device counter: refused, not silently mis-stepped
    pe, ge = _groups(1)
    oe = GomAdam(ge, capturable=True)
    _set_grads(pe, 0)
    held = pe[0].grad
    pe[0].grad = None
    oe.step()
    pe[0].grad = held
    with pytest.raises(RuntimeError):
        oe.step()


def test_gom_adam_refuses_what_it_does_not_implement():
    from gomavatar_amd.optim import GomAdam
    p = torch.zeros(4, device="cuda", requires_grad=True)
    with pytest.raises(NotImplementedError):
        GomAdam([p], weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        GomAdam([p], amsgrad=True)
    q = torch.zeros(4, requires_grad=True)
    q.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        GomAdam([q]).step()
