"""Data-parallel plumbing on CPU (gloo, world_size 2): the bucketed gradient all-reduce of
mslesions3d_amd/parallel.py over the flat gradient arena — coverage, trigger order, averaging."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_buckets, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mslesions3d_amd.engine import ParamArena
        from mslesions3d_amd.parallel import GradBucketReducer, broadcast_model
        from mslesions3d_amd.ssd3d import LSSD3D
        torch.manual_seed(100 + rank)  # different initial weights per rank on purpose
        model = LSSD3D(n_classes=2, input_channels=1, input_size=(64, 64, 64), threshold=[0.1, 0.2])
        arena = ParamArena(model, torch.device("cpu"))
        model._engine.arena = arena
        broadcast_model(model, src=0)
        ref = [torch.zeros_like(arena.flat) for _ in range(world)]
        dist.all_gather(ref, arena.flat)
        assert torch.equal(ref[0], ref[1]), "broadcast must make the replicas identical"
        red = GradBucketReducer(arena, n_buckets=n_buckets)
        assert red.world == world
        # buckets tile the gradient arena exactly once
        assert red.ranges[0][0] == 0 and red.ranges[-1][1] == arena.n_trainable
        assert all(a[1] == b[0] for a, b in zip(red.ranges, red.ranges[1:]))
        assert sorted(k for ks in red.trigger.values() for k in ks) == list(range(len(red.ranges)))
        assert red.stages == set(red.trigger.keys())
        # fake backward: every parameter's gradient becomes available at its stage
        g = torch.Generator().manual_seed(7 + rank)
        local = torch.randn(arena.n_trainable, generator=g)
        arena.grad.zero_()
        fired = []
        for stage in ["heads", 7, 6, 5, 4, 3, 2, 1, 0]:
            for name in arena.names:
                if name in arena.no_grad_names:
                    continue
                st = "heads" if name.startswith("pred_convs") else int(name.split(".")[2])
                if st == stage:
                    lo, n = arena.offsets[name]
                    arena.grad[lo:lo + n] = local[lo:lo + n]
            before = len(red.pending)
            red.on_stage(stage)
            fired.append(len(red.pending) - before)
        scale = red.finish()
        assert scale == 1.0 / world and sum(fired) == len(red.ranges)
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert torch.allclose(arena.grad, both[0] + both[1], rtol=0, atol=0), "bucketed all-reduce must equal the plain sum"
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_buckets", [1, 3])
def test_bucketed_allreduce_gloo_world2(tmp_path, n_buckets):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_buckets, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("n_buckets", [2, 3, 4])
def test_bucket_policy_keeps_the_exposed_tail_small(n_buckets):
    """Only the last bucket's exchange sits in front of the optimiser: it must be the small tail of the gradient (first
    blocks + stem), complete at stage 0, and every earlier bucket must complete at an earlier backward stage."""
    from mslesions3d_amd.engine import ParamArena
    from mslesions3d_amd.parallel import GradBucketReducer
    from mslesions3d_amd.ssd3d import LSSD3D
    model = LSSD3D(n_classes=2, input_channels=1, input_size=(64, 64, 64), threshold=[0.1, 0.2])
    arena = ParamArena(model, torch.device("cpu"))
    red = GradBucketReducer(arena, n_buckets=n_buckets)
    assert not red.active and red.world == 1 and red.stages == set()
    assert len(red.ranges) == n_buckets
    lo, hi = red.ranges[-1]
    assert 0 < hi - lo <= 0.05 * arena.n_trainable
    stage_of = {k: s for s, ks in red.trigger.items() for k in ks}
    assert stage_of[len(red.ranges) - 1] == 0
    order = ["heads", 7, 6, 5, 4, 3, 2, 1, 0]
    pos = [order.index(stage_of[k]) for k in range(len(red.ranges))]
    assert pos == sorted(pos) and pos[-2] < pos[-1]


def _gather_worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mslesions3d_amd.datasets import ShardSampler
        from mslesions3d_amd.predict import gather_detections
        from mslesions3d_amd.train import _mean_over_ranks
        n = 7  # odd: the wrap-around padding makes one subject appear on two ranks
        mine = ShardSampler(n, rank, world, shuffle=False).indices().tolist()
        records = [(pos, f"{pos:04d}", {"boxes": [[0.1 * pos] * 6], "rank": rank}) for pos in mine]
        merged = gather_detections(records, world, rank)
        if rank == 0:
            assert [r[0] for r in merged] == list(range(n)) and [r[1] for r in merged] == [f"{i:04d}" for i in range(n)]
            assert {r[2]["rank"] for r in merged} == {0, 1}
        else:
            assert merged is None
        # validation averages: rank r contributes r + 1 batches whose losses are all (r + 1): mean = (1*1 + 2*2) / 3
        avg = _mean_over_ranks([float((rank + 1) * (rank + 1)), 0.5 * (rank + 1)], rank + 1, torch.device("cpu"), True)
        assert abs(avg[0] - 5.0 / 3.0) < 1e-12 and abs(avg[1] - 0.5) < 1e-12
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_predict_gather_and_validation_mean_gloo_world2(tmp_path):
    """inference is replicas-only with the detections gathered on rank 0; train.py's validation averages are one
    all-reduce so that every rank takes the same early-stopping / checkpoint decisions."""
    world = 2
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
