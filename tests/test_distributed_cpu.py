"""CPU, world_size 2 over gloo: the N>1 path of the harness (vlfm_amd/distributed.py) -- env sharding, MAX/SUM metric
all-reduces -- exercised with the oracle as the per-environment worker (the HIP maps need a GPU; the collective layer does
not).  Sharded result must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _episode_checksum(env_id: int, steps: int) -> float:
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, SyntheticEnv, camera_intrinsics

    fov = camera_intrinsics(160)[2]
    env = SyntheticEnv(env_id, 120, 160)
    vm = RefValueMap(1, use_max_confidence=False)
    for _ in range(steps):
        depth, tf, values = env.observe()
        vm.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, fov)
    return float(vm._value_map.sum() + vm._map.sum())


def _worker(rank: int, world_size: int, port: int, envs_per_rank: int, steps: int, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from vlfm_amd import distributed as D

    D.init("gloo")
    ids = D.shard_env_ids(rank, world_size, envs_per_rank)
    assert all(D.owner_of(e, envs_per_rank) == rank for e in ids)
    local = sum(_episode_checksum(e, steps) for e in ids)
    D.barrier()
    elapsed, (env_steps, checksum, id_sum) = D.reduce_metrics(0.5 + rank, [len(ids) * steps, local, sum(ids)], "cpu")
    out[rank] = (elapsed, env_steps, checksum, id_sum, ids)
    D.shutdown()


def test_sharded_episodes_equal_single_process():
    sys.path.insert(0, ROOT)
    world_size, envs_per_rank, steps = 2, 2, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world_size, port, envs_per_rank, steps, out)) for r in range(world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    want = sum(_episode_checksum(e, steps) for e in range(world_size * envs_per_rank))
    ids_all = sorted(i for r in range(world_size) for i in out[r][4])
    assert ids_all == list(range(world_size * envs_per_rank))          # disjoint cover
    for r in range(world_size):
        elapsed, env_steps, checksum, id_sum, _ = out[r]
        assert elapsed == 0.5 + (world_size - 1)                         # MAX over ranks
        assert env_steps == world_size * envs_per_rank * steps          # SUM over ranks
        assert id_sum == sum(range(world_size * envs_per_rank))
        assert np.isclose(checksum, want, rtol=0, atol=1e-9 * abs(want))


def test_shard_helpers():
    from vlfm_amd import distributed as D

    assert D.shard_env_ids(3, 8, 16) == list(range(48, 64))
    assert [D.owner_of(e, 16) for e in (0, 15, 16, 127)] == [0, 0, 1, 7]
    t, v = D.reduce_metrics(1.25, [3.0, 4.0], "cpu")                     # single process: identity
    assert t == 1.25 and v == [3.0, 4.0]


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no torchrun environment must re-execute itself under torch.distributed.run with two
    ranks (not silently run one): the dry run goes through exactly that launcher, the process-group init, the barriers and
    the MAX/SUM metric all-reduces of the real benchmark, with gloo standing in for RCCL on this CPU-only box."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3",
                        "--envs", "4"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["dry_run"] is True
    assert out["global_envs"] == 8 and out["env_id_checksum"] == sum(range(8))   # both shards took part in the SUM


def test_bench_refuses_a_rank_count_that_does_not_match_gpus():
    import subprocess

    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999",
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
