#!/usr/bin/env python3
"""The host-buffer call (freddy_gpu_ivfadc_search: what pg/freddy_srf.c makes) under the round-6 options, alternating on ONE box:
coarse_pieces (the cell-selection /
table launch per staged piece) x pipeline_batch, at 1024 / 2048 / 4096 / 8192 queries per call.   usage: host_abi_ab.py [reps]"""
import itertools, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
N = 3_000_000
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"])
rng = np.random.default_rng(7)
qid = np.sort(rng.choice(np.arange(1, N + 1), size=8192, replace=False))
hq = x[torch.from_numpy(qid - 1).to(dev)].cpu().numpy()
del x
torch.cuda.empty_cache()
print(f"# GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}  reps={reps}", flush=True)
ref = {}
for rnd in range(2):
    for Q in (1024, 2048, 4096, 8192):
        batches = (512, 2048) if Q <= 1024 else (1024, 2048) if Q <= 4096 else (1024, 2048)
        for pb, cp in itertools.product(batches, (0, 1)):
            index.set_option("pipeline_batch", pb); index.set_option("coarse_pieces", cp)
            gi, gd = index.search(hq[:Q], 5, 10)
            if Q not in ref:
                ref[Q] = (gi.copy(), gd.copy())
            same = np.array_equal(gi, ref[Q][0]) and np.array_equal(gd.view(np.uint32), ref[Q][1].view(np.uint32))
            n = max(8, reps * 1024 // Q)
            for _ in range(3):
                index.search(hq[:Q], 5, 10)
            t0 = time.perf_counter()
            for _ in range(n):
                index.search(hq[:Q], 5, 10)
            dt = (time.perf_counter() - t0) / n
            print(f"round {rnd} Q={Q:5d} pipeline_batch={pb:5d} coarse_pieces={cp}: {dt * 1e3:7.4f} ms  {Q / dt / 1e6:6.3f} M q/s  same_lists={same}", flush=True)
