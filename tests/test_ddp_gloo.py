"""CPU, world_size 2 over gloo: the data-parallel path (batch sharding, DDP gradient averaging,
reduce_dict / gather_all, max-over-ranks timing).  The model is the CPU oracle's Transformer (test
infrastructure) because HIP kernels cannot run here; what is under test is gta_amd.ddp."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from gta_amd import ddp
    from oracle import gta_oracle as O
    r, w, lr = ddp.init_ddp("gloo")
    assert (r, w) == (rank, world)
    f_dims = {"se3": 8, "so2": 8}
    ak = {"f_dims": f_dims, "so2": 2, "so3": 0, "max_freq_h": 1, "max_freq_w": 1}
    aa = {"method": {"name": "gta", "args": ak}}
    torch.manual_seed(0)
    model = O.OracleTransformer(32, 1, 2, 16, 64, 0.0, True, None, False, aa).double()
    g = torch.Generator().manual_seed(1)
    B = 4
    batch = {"x": torch.randn(B, 12, 32, generator=g, dtype=torch.float64),
             "input_transforms": O.random_extrinsics(B, 2, g, torch.float64),
             "input_coord": torch.rand(B, 2, 6, 2, generator=g, dtype=torch.float64),
             "target": torch.randn(B, 12, 32, generator=g, dtype=torch.float64)}

    def loss_of(m, bt):
        reps = O.encoder_reps(ak, bt)
        return ((m(bt["x"], None, reps) - bt["target"]) ** 2).mean()

    # single-process reference gradient on the GLOBAL batch
    ref = {n: torch.autograd.grad(loss_of(model, batch), p, retain_graph=False)[0] for n, p in model.named_parameters()}
    dmodel, blog = ddp.wrap_ddp_logged(model)          # DDP + the bucket timeline hook (bench.py --model-train-steps)
    local = ddp.shard_batch(batch, rank, world)
    assert local["x"].shape[0] == B // world
    loss = loss_of(dmodel, local)
    loss.backward()
    err = max((p.grad - ref[n]).abs().max().item() for n, p in model.named_parameters())
    summ = blog.summary(1)
    assert summ["buckets_per_step"] >= 1 and summ["bytes_per_step"] == sum(p.numel() * 8 for p in model.parameters())
    red = ddp.reduce_dict({"loss": loss.detach().reshape(1), "rank": torch.tensor([float(rank)])})
    ga = ddp.gather_all(torch.tensor([rank * 10, rank * 10 + 1]))
    tmax = ddp.max_over_ranks(1.0 + rank)
    q.put((rank, err, float(red["rank"]), ga.tolist(), tmax, float(red["loss"])))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_data_parallel():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, err, mean_rank, ga, tmax, loss in res:
        assert err < 1e-10            # DDP-averaged per-rank grads == gradient of the global-batch mean loss
        assert abs(mean_rank - 0.5) < 1e-12
        assert ga == [0, 1, 10, 11]
        assert tmax == 2.0
    assert abs(res[0][5] - res[1][5]) < 1e-12     # reduced loss identical on both ranks


def test_shard_batch_rejects_uneven():
    from gta_amd import ddp
    with pytest.raises(ValueError):
        ddp.shard_batch({"x": torch.zeros(5, 3)}, 0, 2)
    assert ddp.shard_batch({"x": torch.arange(6).reshape(6, 1)}, 1, 3)["x"].flatten().tolist() == [2, 3]
    assert ddp.init_ddp() == (0, 1, 0) or "WORLD_SIZE" in os.environ


def test_bench_launch_contract_dry_run_world2():
    """bench.py under the driver's own launch line (torch.distributed.run, --nproc-per-node 2, 127.0.0.1 rendezvous) in
    --dry-run mode: the rendezvous from the environment, both barriers, the MAX over ranks, the per-rank gather and the
    single JSON line on rank 0 run for real (gloo, CPU); only the step is a host no-op.  The first real multi-GPU run
    must not fail on plumbing."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--dry-run"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                       # exactly one JSON line (rank 0 only)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["dry_run"]
    assert len(rec["per_rank_ms_per_step"]) == 2
    # MAX over ranks: rank 1 sleeps twice as long per step as rank 0
    assert rec["ms_per_step"] >= max(rec["per_rank_ms_per_step"]) * 0.95
    assert rec["per_rank_ms_per_step"][1] > rec["per_rank_ms_per_step"][0]
    assert rec["config"]["global_batch"] == 64 and rec["config"]["parallelism"] == "dp2"
    # launched on more than one rank without --model-train-steps, the line carries the whole-model leg: optimizer steps under the two
    # DDP instances of train.py:182-188 and the gradient all-reduce's bucket timeline (the north star's one collective)
    st = rec["srt_train"]
    assert st["steps"] == 5 and st["ms_per_step"] > 0 and st["scenes_per_s"] > 0 and st["ddp"].startswith("two DistributedDataParallel")
    assert set(st["grad_allreduce"]) == {"encoder", "decoder"}
    enc = st["grad_allreduce"]["encoder"]
    assert enc["buckets_per_step"] >= 1 and enc["bytes_per_step"] == (16 * 32 + 32 + 32 * 16 + 16) * 4
