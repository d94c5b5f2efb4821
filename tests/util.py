"""Shared builders of small seeded tables for the parity tests (inputs only -- both the
oracle and the HIP path read exactly these arrays)."""
import functools

import numpy as np
import torch

from freddy_amd import index_build as ib


@functools.lru_cache(maxsize=None)
def corpus(N=20000, d=300, seed=11, dup_frac=0.01):
    torch.manual_seed(0)
    return ib.make_corpus(N, d=d, seed=seed, n_clusters=200, dup_frac=dup_frac, device="cpu")


@functools.lru_cache(maxsize=None)
def ivf_tables(N=20000, C=32, m=12, K=256, seed=5):
    x = corpus(N)
    return ib.build_ivf_index(x, C=C, m=m, K=K, train_size=min(N, 5000), iters=4, seed=seed)


@functools.lru_cache(maxsize=None)
def pq_tables(N=20000, m=12, K=256, seed=6):
    x = corpus(N)
    return ib.build_pq_index(x, m=m, K=K, train_size=min(N, 5000), iters=4, seed=seed)


@functools.lru_cache(maxsize=None)
def ivpq_tables(N=20000, m=30, K=32, k_coarse=8, seed=7):
    x = corpus(N)
    return ib.build_ivpq_index(x, m=m, K=K, k_coarse=k_coarse, train_size=min(N, 5000), iters=4, seed=seed)


def queries_from_corpus(N, Q, seed=7):
    rng = np.random.default_rng(seed)
    ids = np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False)).astype(np.int32)
    x = corpus(N)
    return ids, x[torch.from_numpy(ids.astype(np.int64) - 1)].numpy().astype(np.float32)


def assert_same_lists(got_ids, got_dist, exp, what=""):
    """bit-exact (id, rank) and bit-exact float distance."""
    exp_ids = exp["id"].reshape(got_ids.shape)
    exp_d = exp["dist"].reshape(got_dist.shape)
    bad = np.argwhere(exp_ids != got_ids)
    assert bad.size == 0, f"{what}: id mismatch at {bad[:5].tolist()} exp {exp_ids[tuple(bad[0])]} got {got_ids[tuple(bad[0])]}"
    assert np.array_equal(exp_d.view(np.uint32), got_dist.view(np.uint32)), f"{what}: distance bits differ"
