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
