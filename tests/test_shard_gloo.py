"""The N>1 path on CPU: two gloo ranks shard a batch through the very function bench.py --gpus 2 calls
(bench.rank_workload -> gama_tts_amd.shard.shard_range) and reduce the elapsed time the way it does."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
from gama_tts_amd.shard import max_over_ranks, shard_range, sum_over_ranks
import tracks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py's own argument parsing and shard computation, with the ranks torch.distributed hands out
    args = bench.parse_args(["--gpus", str(world), "--global-batch", str(total), "--dist-backend", "gloo"])
    g_total, lo, hi = bench.rank_workload(args, dist.get_rank(), dist.get_world_size())
    assert g_total == total and (lo, hi) == shard_range(total, rank, world)
    # every rank builds only its own utterances; seeds are global utterance ids
    mine = np.stack([tracks.random_track(6, 1000 + b) for b in range(lo, hi)]) if hi > lo else np.zeros((0, 6, 16), np.float32)
    elapsed = 1.0 + rank  # the slower rank defines the step time
    slowest = max_over_ranks(elapsed, dist)
    n_total = sum_over_ranks(hi - lo, dist)
    dist.barrier()
    np.save(os.path.join(out_dir, "r%d.npy" % rank), mine)
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write("%d %d %.1f %.1f" % (lo, hi, slowest, n_total))
    dist.destroy_process_group()


def test_two_ranks_partition_the_batch(tmp_path):
    total, world = 7, 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    parts, covered = [], []
    for r in range(world):
        lo, hi, slowest, n_total = open(tmp_path / ("r%d.txt" % r)).read().split()
        assert float(slowest) == 2.0 and float(n_total) == total
        covered += list(range(int(lo), int(hi)))
        parts.append(np.load(tmp_path / ("r%d.npy" % r)))
    assert covered == list(range(total))
    whole = np.stack([tracks.random_track(6, 1000 + b) for b in range(total)])
    assert np.array_equal(np.concatenate(parts), whole)


def test_shard_range_properties():
    for total in (0, 1, 5, 256, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_rank_workload_defaults_match_baseline_configs():
    # N = 1: configs[3] (4096 x 7500 frames, SectionDelay 2); N = 8: configs[4] (4096 per GPU, 32768 in all)
    a = bench.parse_args([])
    assert (a.frames, a.delay, a.precision) == (7500, 2, "f32")
    assert bench.rank_workload(a, 0, 1) == (4096, 0, 4096)
    a8 = bench.parse_args(["--gpus", "8"])
    spans = [bench.rank_workload(a8, r, 8) for r in range(8)]
    assert all(t == 32768 for t, _, _ in spans)
    assert [hi - lo for _, lo, hi in spans] == [4096] * 8 and spans[0][1] == 0 and spans[-1][2] == 32768
    # a fixed global batch that does not divide: contiguous, sizes differ by at most one
    ag = bench.parse_args(["--gpus", "3", "--global-batch", "1000"])
    sizes = [bench.rank_workload(ag, r, 3)[2] - bench.rank_workload(ag, r, 3)[1] for r in range(3)]
    assert sum(sizes) == 1000 and max(sizes) - min(sizes) <= 1
    # the other tubes keep the 256 x 2 s workload
    a5 = bench.parse_args(["--model", "5"])
    assert (a5.batch, a5.frames, a5.delay, a5.precision, a5.output_rate) == (256, 500, 1, "f64", 48000.0)


def test_launch_plan_is_what_a_launcher_would_export():
    """`python bench.py --gpus N` without torch.distributed.run: one child per device with the launcher's environment."""
    plan = bench.launch_plan(4, ["--gpus", "4", "--steps", "3"], 29123)
    assert len(plan) == 4
    for r, (cmd, env) in enumerate(plan):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
        assert env == {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "4", "LOCAL_WORLD_SIZE": "4",
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29123"}


def _run_bench(argv, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_starts_its_own_ranks_from_a_plain_python_start():
    """A plain `python bench.py --gpus 2` (no WORLD_SIZE): the parent spawns two fresh rank processes before touching any
    GPU, they rendezvous on 127.0.0.1 and cut the batch into contiguous shards; ONE line comes back, from rank 0, with
    n_gpus = 2.  (--rehearse-launch: the launcher, the process group and the rank logic without the GPU work, so that
    this runs on the CPU-only build box; BASELINE configs[4] = 4096 utterances per GPU.)"""
    r, line = _run_bench(["--gpus", "2", "--rehearse-launch"])
    assert r.returncode == 0, r.stderr
    assert line == {"rehearsal": True, "n_gpus": 2, "global_batch": 8192, "shards": [[0, 4096], [4096, 8192]],
                    "max_over_ranks_of_1_plus_rank": 2.0, "launcher": "self"}
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1


def test_bench_self_launch_with_a_fixed_global_batch_and_a_failing_rank():
    r, line = _run_bench(["--gpus", "3", "--global-batch", "1000", "--rehearse-launch"])
    assert r.returncode == 0, r.stderr
    assert line["n_gpus"] == 3 and line["shards"] == [[0, 334], [334, 667], [667, 1000]]
    # without GPUs the real (non-rehearsal) ranks fail loudly, and the parent reports it instead of hanging in a barrier
    import torch
    if not torch.cuda.is_available():
        r, line = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
        assert r.returncode != 0 and line is None
        assert "needs a GPU" in r.stderr
