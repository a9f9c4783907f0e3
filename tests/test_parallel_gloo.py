"""World-size-2 test of the clip sharding path on CPU (gloo).  The model call is replaced by a pure function of the
batch: what is under test is coverage (every clip exactly once, original order) and the gather."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lip2speech_amd import parallel


def _worker(rank, size, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    batches = [torch.arange(i * 4, i * 4 + 4, dtype=torch.float32) for i in range(7)]      # 7 batches of 4 "clips"
    out = parallel.run_sharded(batches, lambda b: (b * 2).tolist())
    mine = parallel.shard_batches(batches)
    ret[rank] = (out, len(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 33):
        for size in (1, 2, 3, 8):
            got = [i for r in range(size) for i in parallel.shard_range(n, r, size)]
            assert got == list(range(n))
            sizes = [len(parallel.shard_range(n, r, size)) for r in range(size)]
            assert max(sizes) - min(sizes) <= 1


def test_run_sharded_world2():
    size = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(size, 29513, ret), nprocs=size, join=True)
        expect = [[float(2 * (i * 4 + j)) for j in range(4)] for i in range(7)]
        for rank in range(size):
            out, n_mine = ret[rank]
            assert out == expect                       # every rank sees all batches, in order, exactly once
            assert n_mine == (4 if rank == 0 else 3)


def test_balanced_groups():
    """`InflightPool.map` cuts K same-shape batches into a multiple-of-chains number of groups whose sizes differ by at most one
    (bench.py's 20-step driver run: four groups of 5 on two chains, not 8 + 8 + 4)."""
    from lip2speech_amd.parallel import InflightPool
    for k, g, c in [(20, 8, 2), (192, 8, 2), (21, 8, 2), (3, 8, 2), (1, 8, 2), (7, 3, 2), (8, 8, 1), (17, 8, 2), (5, 8, 4)]:
        groups = InflightPool.balanced_groups(list(range(k)), g, c)
        sizes = [len(x) for x in groups]
        assert [i for x in groups for i in x] == list(range(k))
        assert max(sizes) <= g and max(sizes) - min(sizes) <= 1
        assert len(groups) % c == 0 or k < c
    assert [len(x) for x in InflightPool.balanced_groups(list(range(20)), 8, 2)] == [5, 5, 5, 5]
    assert [len(x) for x in InflightPool.balanced_groups(list(range(20)), 8, 3)] == [7, 7, 6]
    # chains in flight by rows per launch: 20 batches fill three chains' launches better than two's, 192 fill two's completely
    assert InflightPool.chains_for(20, 8) == 3 and InflightPool.chains_for(192, 8) == 3 and InflightPool.chains_for(16, 8) == 2
    assert InflightPool.chains_for(1, 8) == 1 and InflightPool.chains_for(5, 1) == 3
