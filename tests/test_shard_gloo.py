"""The N>1 path on CPU: two gloo ranks shard a batch the way bench.py/--gpus 2 does."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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
    lo, hi = shard_range(total, rank, world)
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
