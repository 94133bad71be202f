"""Data-parallel parity on the GPU, as SURVEY.md §8(e) defines it: for each rank r the loss and the LOCAL gradients equal
the CPU oracle on shard r; the reduced gradient equals the mean of the per-shard oracle gradients (it intentionally does
NOT equal one batch-2N run: BatchNorm statistics and the loss normaliser are per replica, as the reference's single-GPU
semantics per shard / DDP).  Two ranks share this one GPU over gloo (RCCL refuses two ranks on one device); the arithmetic
and the bucket / stream plumbing are the ones the RCCL run uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GRAD_TOL = 2e-3   # the gradient bar of tests/test_gpu_model.py (BASELINE.md "Gradient tolerance")
LOSS_TOL = 1e-4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mslesions3d_amd.parallel import GradBucketReducer, broadcast_model
        from mslesions3d_amd.ssd3d import LSSD3D, MultiBoxLoss
        from mslesions3d_amd.trainer import FusedTrainer
        from oracle.multibox import multibox_loss
        from tests.golden import detinit
        from tests.util import oracle_model
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        size, n = (64, 64, 64), 2
        m = LSSD3D(n_classes=2, input_channels=1, input_size=size, threshold=[0.1, 0.2], lr=1e-3)
        m.load_state_dict(detinit.fill_state_dict(m.state_dict(), 1234 + 5 * rank))  # rank 1 starts different on purpose
        m = m.to(dev).train()
        m._ensure_device_state(dev)
        eng = m._engine
        arena = eng.ensure_arena(dev)
        broadcast_model(m, src=0)  # -> everybody holds rank 0's weights (seed 1234) and BatchNorm buffers

        # per-shard oracle (CPU): every rank computes all shards' gradients, so it can form their mean itself
        oracle_grads, oracle_loss = [], []
        for r in range(world):
            o = oracle_model(1, size).train()
            xs = detinit.make_volume_batch(50 + r, n, 1, size)
            bs, ls = detinit.make_gt(60 + r, n, size)
            lo, sc = o(xs)
            cf, lc = multibox_loss(lo, sc, bs, ls, o.priors_cxcycz, [0.1, 0.2])
            (cf + lc).backward()
            oracle_grads.append({k: p.grad for k, p in o.named_parameters() if p.grad is not None})
            oracle_loss.append((cf.item(), lc.item()))

        # this rank's shard through the HIP path: local gradients first (no exchange)
        xs = detinit.make_volume_batch(50 + rank, n, 1, size).to(dev)
        bs, ls = detinit.make_gt(60 + rank, n, size)
        gb, gl, off, T = MultiBoxLoss.pack_targets(bs, ls, dev)
        tr = FusedTrainer(m)
        lf = m.loss_fn
        locs, scores = eng.forward(xs, training=True, need_grad=True, nan_check=False)
        pl = eng.plan_for(xs, True)
        st = lf._state(n, m.priors_cxcycz.shape[0], 2, T, dev)
        up = torch.tensor([1.0, 1.0], dtype=torch.float32, device=dev)
        lf._run_forward(st, locs, scores, gb, gl, off, T, with_backward_upstream=up, nan_flag=pl.nan_flag)
        eng.backward(pl, st["dlocs"], st["dscores"])
        torch.cuda.synchronize()
        conf, loc, _ = st["loss_out"].tolist()
        assert abs(conf - oracle_loss[rank][0]) <= LOSS_TOL * abs(oracle_loss[rank][0])
        assert abs(loc - oracle_loss[rank][1]) <= LOSS_TOL * abs(oracle_loss[rank][1])
        worst = max((_relerr(arena.grad_views[k], g), k) for k, g in oracle_grads[rank].items())
        assert worst[0] <= GRAD_TOL, f"rank {rank}: local gradient vs the oracle on its shard: {worst}"
        local = arena.grad.clone()

        # the same backward with the bucketed exchange hooked in: SUM over ranks, the mean is folded into Adam (scale)
        red = GradBucketReducer(arena, n_buckets=3)
        assert red.active and red.world == world
        eng.forward(xs, training=True, need_grad=True, nan_check=False)
        lf._run_forward(st, locs, scores, gb, gl, off, T, with_backward_upstream=up, nan_flag=pl.nan_flag)
        eng.backward(pl, st["dlocs"], st["dscores"], on_bucket_ready=red)
        scale = red.finish()
        torch.cuda.synchronize()
        assert scale == 1.0 / world
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert torch.equal(arena.grad, both[0] + both[1]), "bucketed exchange must equal the plain sum of the local gradients"
        mean = {k: sum(g[k] for g in oracle_grads) / world for k in oracle_grads[0]}
        worst = max((_relerr(arena.grad_views[k] * scale, g), k) for k, g in mean.items())
        assert worst[0] <= GRAD_TOL, f"reduced gradient vs the mean of the per-shard oracle gradients: {worst}"

        # and the full fused step (exchange overlapped, Adam with the 1/world scale) keeps the replicas identical
        for s in range(3):
            tr.step_packed(xs, gb, gl, off, T, sync=False)
        torch.cuda.synchronize()
        flats = [torch.zeros_like(arena.flat) for _ in range(world)]
        dist.all_gather(flats, arena.flat)
        assert torch.equal(flats[0], flats[1]), "replicas diverged"
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_per_shard_parity_and_reduced_gradient_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
