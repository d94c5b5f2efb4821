"""world_size-2 gloo test of the multi-GPU host logic (query sharding + gather of per-shard
top-k).  The per-rank search is stood in by the oracle -- this test is about the distributed
plumbing, which is identical for RCCL on the GPU box."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util


def _worker(rank, world, port, Q, k, W, ret):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from freddy_amd import shard
    from oracle.oracle import Oracle
    o = Oracle()
    t = util.ivf_tables(N=3000, C=12, K=64)
    ot = o.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(3000, Q)

    def search(local):
        r = o.ivfadc_search_many(ot, local.numpy(), k, W)
        return torch.from_numpy(r["id"].copy()), torch.from_numpy(r["dist"].copy())

    ids, dd = shard.sharded_search(search, torch.from_numpy(qs), k)
    if rank == 0:
        ret["ids"], ret["dist"] = ids.numpy(), dd.numpy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("Q", [9, 16])
def test_two_rank_sharded_search_equals_single_process(oracle, Q):
    k, W = 5, 3
    t = util.ivf_tables(N=3000, C=12, K=64)
    ot = oracle.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    _, qs = util.queries_from_corpus(3000, Q)
    exp = oracle.ivfadc_search_many(ot, qs, k, W)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + os.getpid() % 300 + Q
    mp.spawn(_worker, args=(2, port, Q, k, W, ret), nprocs=2, join=True)
    assert np.array_equal(ret["ids"], exp["id"])
    assert np.array_equal(ret["dist"].view(np.uint32), exp["dist"].view(np.uint32))


def test_shard_bounds_cover_everything():
    from freddy_amd import shard
    for Q in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_bounds(Q, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == Q
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
