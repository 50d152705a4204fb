"""Frame-parallel path on CPU: 2 processes over gloo.  The HIP kernels cannot run
here, so each rank produces its per-frame gradient from a deterministic stand-in
of the frame step; what is checked is the distributed logic itself: frame
assignment, the single flat all-reduce, identical parameters on every rank, and
equivalence with one process that averages the same frames."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gomavatar_amd.parallel import FrameParallel, shapes_for_model

N, F = 50, 96
STEPS = 3


def fake_frame_grads(params, frame):
    """Deterministic function of (parameters, frame index) standing in for RenderStep."""
    g = torch.Generator().manual_seed(1234 + frame)
    return {k: torch.randn(v.shape, generator=g) * 0.1 + 0.01 * v.detach() for k, v in params.items()}


def run_steps(fp, world_frames):
    opt = fp.make_adam({"default": 1e-2, "vertices": 1e-3})
    for step in range(STEPS):
        frames = world_frames(step)
        fp.grads.flat.zero_()
        for fr in frames:
            for k, g in fake_frame_grads(fp.params.views, fr).items():
                fp.grads[k].add_(g)
        if len(frames) > 1:
            fp.grads.flat.mul_(1.0 / len(frames))
        fp.all_reduce_grads()
        opt.step()
    return fp.params.flat.detach().clone()


def _init_params(fp):
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        fp.params.flat.copy_(torch.randn(fp.params.numel, generator=g))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fp = FrameParallel(shapes_for_model(N, F), "cpu", pad_to=N * 3 + F * 9 + 17)
        assert fp.world == world and fp.rank == rank
        if rank == 0:
            _init_params(fp)
        fp.broadcast_params(0)
        res = run_steps(fp, lambda step: [fp.frame_index(step)])
        gathered = [torch.zeros_like(res) for _ in range(world)]
        dist.all_gather(gathered, res)
        if rank == 0:
            assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
            torch.save(gathered[0], out)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(120)
def test_two_ranks_match_single_process_average(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    dp = torch.load(out)
    fp = FrameParallel(shapes_for_model(N, F), "cpu")
    _init_params(fp)
    ref = run_steps(fp, lambda step: [2 * step, 2 * step + 1])   # the same frames, averaged in one process
    assert torch.allclose(dp, ref, rtol=1e-6, atol=1e-7)


def test_flat_buffer_views_alias_the_payload():
    fp = FrameParallel(shapes_for_model(N, F), "cpu", pad_to=1000)
    assert fp.grads.numel == max(1000, 3 * N + 9 * F)
    fp.grads["so3"].fill_(2.0)
    assert float(fp.grads.flat.sum()) == 2.0 * 3 * F
    assert fp.grads["vertices"].shape == (3, N) and fp.grads["appearance"].data_ptr() == fp.grads.flat[3 * N + 6 * F:].data_ptr()


def _recover_worker(rank, world, port, out):
    """Round-5 advisor finding: after ZeRO-1 steps a rank holds current moments (and, after a timed-out gather, current parameters) for ITS OWN slice
    only; recover() must take every slice from its owner, not broadcast rank 0's stale copy of the others.  The sharded optimizer is a stand-in with
    FlatAdam's fields (the HIP exchange cannot run here); zero1_slice / gather_optimizer_state / recover are the real host logic."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fp = FrameParallel(shapes_for_model(N, F), "cpu", align=4)
        n = fp.params.numel
        g = torch.Generator().manual_seed(7)
        truth = {k: torch.randn(n, generator=g) for k in ("params", "m", "v")}           # what every slice's OWNER holds
        lo, hi = fp.zero1_slice()
        stale = lambda t: torch.full_like(t, float(100 + rank))                             # what a rank holds of the slices it does not own
        class Opt: pass
        opt = Opt()
        opt.exp_avg, opt.exp_avg_sq, opt.t, opt.moments_sharded = stale(truth["m"]), stale(truth["v"]), 10 + rank, True
        fp.params.flat.copy_(stale(truth["params"]))
        for dst, src in ((fp.params.flat, truth["params"]), (opt.exp_avg, truth["m"]), (opt.exp_avg_sq, truth["v"])):
            dst[lo:hi] = src[lo:hi]
        fp.recover(opt)
        ok = (torch.equal(fp.params.flat, truth["params"]) and torch.equal(opt.exp_avg, truth["m"]) and torch.equal(opt.exp_avg_sq, truth["v"])
              and opt.t == 10 and opt.moments_sharded is False)
        # a REPLICATED optimizer (collective / peer) still takes everything from rank `src`
        opt.exp_avg, opt.exp_avg_sq, opt.t, opt.moments_sharded = stale(truth["m"]), stale(truth["v"]), 20 + rank, False
        fp.params.flat.copy_(stale(truth["params"]))
        fp.recover(opt, src=1)
        ok = ok and bool((fp.params.flat == 101).all() and (opt.exp_avg == 101).all() and (opt.exp_avg_sq == 101).all()) and opt.t == 21
        res = [None] * world
        dist.all_gather_object(res, ok)
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_recover_takes_each_zero1_slice_from_its_owner(tmp_path):
    out = str(tmp_path / "rec.pt")
    mp.spawn(_recover_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out, weights_only=False) == [True, True]
