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


def _worker(rank, world, port, result_dir, dtype="f32"):
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
        m.compute_dtype = dtype
        bf16 = dtype == "bf16"
        # bf16 activations (a build-side extension): the fp32 oracle is the yard stick only for the losses (2e-2, the bar of
        # tests/test_gpu_bf16.py); the exchange itself is checked exactly (sum of the local gradients, replicas identical)
        loss_tol, grad_tol = (2e-2, None) if bf16 else (LOSS_TOL, GRAD_TOL)
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
        assert abs(conf - oracle_loss[rank][0]) <= loss_tol * abs(oracle_loss[rank][0])
        assert abs(loc - oracle_loss[rank][1]) <= loss_tol * abs(oracle_loss[rank][1])
        if grad_tol is not None:
            worst = max((_relerr(arena.grad_views[k], g), k) for k, g in oracle_grads[rank].items())
            assert worst[0] <= grad_tol, f"rank {rank}: local gradient vs the oracle on its shard: {worst}"
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
        # every bucket was handed over while the backward pass was still being enqueued (bf16 too): all stages fired
        assert sorted(red.trigger.keys(), key=str) == sorted(red.stages, key=str) and len(red.ranges) == 3
        if grad_tol is not None:
            mean = {k: sum(g[k] for g in oracle_grads) / world for k in oracle_grads[0]}
            worst = max((_relerr(arena.grad_views[k] * scale, g), k) for k, g in mean.items())
            assert worst[0] <= grad_tol, f"reduced gradient vs the mean of the per-shard oracle gradients: {worst}"

        # and the full fused step (exchange overlapped, Adam with the 1/world scale) keeps the replicas identical
        for s in range(3):
            tr.step_packed(xs, gb, gl, off, T, sync=False)
        torch.cuda.synchronize()
        flats = [torch.zeros_like(arena.flat) for _ in range(world)]
        dist.all_gather(flats, arena.flat)
        assert torch.equal(flats[0], flats[1]), "replicas diverged"
        # the same trainer / reducer switched to the other activation dtype (state left by an fp32 step must
        # not make a bf16 step skip the wait of the communication stream): replicas stay identical
        m.compute_dtype = "f32" if bf16 else "bf16"
        for s in range(3):
            tr.step_packed(xs, gb, gl, off, T, sync=False)
        torch.cuda.synchronize()
        dist.all_gather(flats, arena.flat)
        assert torch.equal(flats[0], flats[1]), "replicas diverged after switching the activation dtype"
        assert bool(torch.isfinite(arena.flat).all())
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_per_shard_parity_and_reduced_gradient_two_ranks(tmp_path, dtype):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), dtype), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_train_and_predict_entry_points_under_torchrun_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m mslesions3d_amd.train` (BASELINE north_star: "data-parallel
    training shards synthetic volumes across the GPUs" through the train.py entry point; loop shape train.py:171-188): the
    two ranks train on disjoint shards of every epoch, end with bit-identical replicas (train.py checks and raises), rank 0
    alone writes metrics and checkpoints; then `-m mslesions3d_amd.predict` with two replicas gathers every subject's
    detections on rank 0 and writes the same files as a single process."""
    import glob
    import json
    import subprocess
    import sys
    from mslesions3d_amd import datasets as DS
    from mslesions3d_amd import predict as P
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    DS.generate_artificial_dataset(str(tmp_path / "data"), "toy64", num_images=10, image_size=(64, 64, 64))
    env = dict(os.environ, MSL_DP_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", PYTHONPATH=root)

    def torchrun(module, args):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "-m", module] + args
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        return r

    logs = tmp_path / "logs"
    torchrun("mslesions3d_amd.train", ["-d", str(tmp_path / "data"), "-dn", "toy64", "-b", "2", "-me", "2", "-ld", str(logs), "-en", "run",
                                       "-lr", "0.001"])
    run = logs / "run"
    lines = [json.loads(l) for l in open(run / "metrics.jsonl")]
    val = [l for l in lines if "avg_val_loss" in l]
    assert len(val) == 2 and all(l["avg_val_loss"] == l["avg_val_loss"] for l in val)
    # 8 training cases over 2 ranks x batch 2 -> 2 steps per epoch and rank
    assert max(l["step"] for l in lines) == 4
    shards = [[json.loads(l) for l in open(run / f"shard_rank{r}.jsonl")] for r in range(2)]
    for epoch in (0, 1):
        seen = [sorted(s for l in shards[r] if l["epoch"] == epoch for s in l["subjects"]) for r in range(2)]
        assert len(seen[0]) == len(seen[1]) == 4 and not (set(seen[0]) & set(seen[1])), seen
        assert len(set(seen[0]) | set(seen[1])) == 8   # together: every training case, once
    ckpts = sorted(glob.glob(str(run / "checkpoint-*.ckpt")))
    assert 1 <= len(ckpts) <= 3 and (run / "last.ckpt").exists()
    # prediction: two replicas + gather on rank 0 == one process
    out2, out1 = tmp_path / "pred2", tmp_path / "pred1"
    common = ["-d", str(tmp_path / "data"), "-dn", "toy64", "-m", ckpts[0], "-ps", "train", "-sc", "0.3"]
    torchrun("mslesions3d_amd.predict", common + ["-o", str(out2)])
    env1 = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "mslesions3d_amd.predict"] + common + ["-o", str(out1)], cwd=root, env=env1,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    names = sorted(os.path.basename(f) for f in glob.glob(str(out1 / "*")))
    assert names == sorted(os.path.basename(f) for f in glob.glob(str(out2 / "*"))) and len([n for n in names if n.endswith(".json")]) == 8 + 2
    for n in names:
        assert open(out1 / n, "rb").read() == open(out2 / n, "rb").read(), n


@pytest.mark.parametrize("mode,extra", [("train", []), ("train", ["--dtype", "bf16"]), ("infer", [])])
def test_bench_under_torchrun_two_ranks(mode, extra):
    """bench.py launched as the driver launches it for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py
    --gpus N --steps K --warmup W`), here with two ranks sharing the one GPU over gloo (MSL_BENCH_BACKEND: functional
    rehearsal; the real run is one rank per GPU over RCCL): the set-up phase, the sampled event pairs, the reducer's hooks
    between program segments and the stop-event forks all run on both ranks, rank 0 prints ONE JSON line whose value is the
    whole-job aggregate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MSL_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--mode", mode] + extra
    if mode == "infer":
        cmd += ["--map-cases", "2", "--train-steps", "2"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    per_step = (4 if mode == "train" else 2) * 2  # volumes of both ranks per step
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["config"]["global_batch"] == per_step
    if mode == "train":
        assert d["config"]["parallelism"] == "dp2" and d["roofline"]["launches_timed"] == 3
