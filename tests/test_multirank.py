"""N>1 path on CPU: two gloo ranks shard frames with no data-path collective; only sizes and the
timing reduction cross ranks.  The descriptor builder is host code, so this runs without a GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import FRAMES, ROOT, read_frame


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zstd_amd import api, sharding
    entries = [e for e in FRAMES if "error" not in e and "dict" not in e and e["capacity"] <= (1 << 20)
               and e["csize"] > 0 and not e["name"].startswith(("concat", "skippable"))]
    costs = [e["csize"] + e["size"] for e in entries]
    mine = sharding.partition_frames(costs, world)[rank]
    frames = [read_frame(entries[i]) for i in mine]
    src, fin, _ = api.pack_frames(frames, [entries[i]["capacity"] for i in mine])
    fds, bds, nb = api.build_descriptors(src, fin)
    assert all(fds[i].hostError == 0 for i in range(len(mine)))
    local = torch.tensor([len(mine), sum(costs[i] for i in mine), nb], dtype=torch.int64)
    total = local.clone()
    dist.all_reduce(total)                      # bookkeeping only: counts and bytes
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    t = sharding.max_over_ranks(dist, 1.0 + rank)
    seeds = list(sharding.shard_seed_range(rank, 4))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, local.tolist(), total.tolist(), gathered, t, seeds, len(entries), sum(costs)))


def test_two_ranks_shard_frames_without_overlap():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
        assert p.exitcode == 0, "a rank failed"
    res = sorted(q.get(timeout=10) for _ in range(world))
    n_entries, cost_sum = res[0][6], res[0][7]
    shards = res[0][3]
    assert sorted(shards[0] + shards[1]) == list(range(n_entries))          # disjoint cover
    assert res[0][2] == res[1][2] and res[0][2][0] == n_entries and res[0][2][1] == cost_sum
    loads = [r[1][1] for r in res]
    assert max(loads) - min(loads) <= max(loads) * 0.2                        # balanced by C+D bytes
    assert res[0][4] == res[1][4] == 2.0                                      # max over ranks
    assert res[0][5] == [0, 1, 2, 3] and res[1][5] == [4, 5, 6, 7]            # weak-scaling seed ranges


def test_partition_is_deterministic_and_balanced():
    from zstd_amd import sharding
    rng = np.random.default_rng(0)
    costs = rng.integers(1000, 200000, size=1000).tolist()
    a = sharding.partition_frames(costs, 8)
    assert a == sharding.partition_frames(costs, 8)
    loads = [sum(costs[i] for i in s) for s in a]
    assert sorted(i for s in a for i in s) == list(range(1000))
    assert max(loads) / min(loads) < 1.02
