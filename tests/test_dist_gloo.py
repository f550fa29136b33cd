"""world_size-2 gloo test of the N>1 bench path on CPU: batch sharding with no data-path collective,
barrier + max-over-ranks timing, whole-job aggregation.  The per-rank 'compute' is the
oracle (test infrastructure) so that shard-vs-single equality of concatenated outputs is checked."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

import taco_oracle as O
from util import tiny_hp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import taco_amd
    from taco_amd import dist as D
    import torch.distributed as dist
    D.init_process_group("gloo")
    hp = tiny_hp()
    w = O.init_weights(hp, 1, 0)
    ids, L = O.synthetic_inputs(5, 8, 11, ragged=True)
    lo, hi = D.shard_range(5, rank, world)
    out = O.forward(w, hp, ids[lo:hi], L[lo:hi])
    dist.barrier()
    t = D.max_over_ranks(1.0 + rank)            # slowest rank wins
    frames = D.sum_over_ranks(out["mel"].shape[0] * out["mel"].shape[1])
    parts = D.gather_rows(out["mel"], world)
    if rank == 0:
        q.put((t, frames, np.concatenate(parts, 0)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_inference_equals_single():
    hp = tiny_hp()
    w = O.init_weights(hp, 1, 0)
    ids, L = O.synthetic_inputs(5, 8, 11, ragged=True)
    ref = O.forward(w, hp, ids, L)["mel"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t, frames, mel = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0
    assert frames == ref.shape[0] * ref.shape[1]
    assert np.allclose(mel, ref, atol=1e-12)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    from taco_amd import dist as D
    from taco_amd.train_ops import allreduce_gradients
    D.init_process_group("gloo")
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)       # per-shard gradient of a per-shard mean loss
    allreduce_gradients(g)
    if rank == 0:
        q.put(g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_is_the_mean_of_the_shards():
    """X1: one all-reduce of the flat bucket, divided by world size = gradient of the global-batch mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(g, np.arange(1000) * 1.5)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as typed (the driver's command form): bench.py re-launches itself under torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1; --selftest-launcher runs exactly that path on gloo (no GPU work) and reports the
    world size the ranks saw and the max-over-ranks reduction the timed region uses."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launcher"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["world_size"] == 2 and rec["n_gpus"] == 2 and rec["max_over_ranks"] == 2.0


def test_bench_strong_and_weak_scaling_shard_the_batch_as_survey_8e_says():
    """SURVEY 8e asks for both: weak (per-GPU batch held at the workload's 32 rows: global batch N x 32) and strong (the workload's batch is the
    GLOBAL batch, split over the ranks).  The launcher self-test reports the sharding rule of the timed path for two ranks."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for scaling, rows, glob in (("weak", 32, 64), ("strong", 16, 32)):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scaling", scaling, "--selftest-launcher"], env=env,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert rec["world_size"] == 2 and rec["scaling"] == scaling and rec["rows_per_gpu"] == rows and rec["global_batch"] == glob, rec


def test_bench_single_process_selftest():
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--selftest-launcher"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["world_size"] == 1


def test_bench_train_spawns_its_own_ranks():
    """tools/bench_train.py --gpus 2 launches its own ranks; the self-test runs the launcher, the gloo rendezvous and the flat
    gradient all-reduce (mean over ranks: element 1 of arange * (rank+1) averages to 1.5)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_train.py"), "--gpus", "2", "--selftest-launcher"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["world_size"] == 2 and abs(rec["grad_mean_factor"] - 1.5) < 1e-6
