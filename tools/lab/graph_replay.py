#!/usr/bin/env python3
"""Experiment (round 6): a batch's chain of launches captured into a hipGraph per (stream, buffer) pair and replayed, against the
plain *_dev calls -- four batches in flight, config 3.  usage: graph_replay.py [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
N, Q, k, W, n_fl = 3_000_000, 1024, 5, 10, 4
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
rng = np.random.default_rng(7)
d_qs = [x[torch.from_numpy(np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False)) - 1).to(dev)].contiguous() for _ in range(n_fl)]
streams = [torch.cuda.Stream(dev) for _ in range(n_fl)]
res = [torch.zeros((2, Q, k), dtype=torch.int32, device=dev) for _ in range(n_fl)]
st_w = torch.zeros(4, dtype=torch.int32, device=dev)
index.set_option("scan_share", n_fl)
calls = [index.bind_search_dev(d_qs[i].data_ptr(), Q, k, W, 1000.0, gpu.FOUND_ROWS, res[i][0].data_ptr(), res[i][1].data_ptr(), st_w.data_ptr(), streams[i].cuda_stream)
         for i in range(n_fl)]
for _ in range(3):
    for c in calls:
        c()
torch.cuda.synchronize(dev)
ref = [r.clone() for r in res]


def timed(fns, n):
    for i in range(8):
        fns[i % n_fl]()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n):
        fns[i % n_fl]()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n, th / n


graphs = []
for i in range(n_fl):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[i]):
        calls[i]()
    graphs.append(g)
torch.cuda.synchronize(dev)
for r in res:
    r.zero_()


def replay_on(i):
    g, st = graphs[i], streams[i]
    def f():
        with torch.cuda.stream(st):
            g.replay()
    return f


rep = [replay_on(i) for i in range(n_fl)]
for f in rep:
    f()
torch.cuda.synchronize(dev)
same = all(torch.equal(a, b) for a, b in zip(res, ref))
for rnd in range(3):
    a, ah = timed(calls, steps)
    b, bh = timed(rep, steps)
    print(f"round {rnd}: plain {a * 1e3:.4f} ms per step ({Q / a / 1e6:.2f} M q/s, host {ah * 1e6:.1f} us)   graph replay {b * 1e3:.4f} ms ({Q / b / 1e6:.2f} M q/s, host {bh * 1e6:.1f} us)   same lists: {same}", flush=True)
