"""GPU-box helper: the synchronous host-buffer call at the metric's batch size (1024 queries) and around it -- how should a call
that is smaller than the pipeline's sub-batch (2048) be cut?  usage: pipe_small.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
N = 3_000_000
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"])
rng = np.random.default_rng(7)
qid = rng.choice(np.arange(1, N + 1), size=4096, replace=False)
hq = x[torch.from_numpy(qid - 1).to(dev)].cpu().numpy()
ref = {}
for Q in (512, 1024, 2048, 4096):
    for lanes in (1, 2, 4):
        for batch in (128, 256, 512, 1024, 2048):
            if batch > Q or (Q // batch) > 16: continue
            index.set_option("pipeline_lanes", lanes); index.set_option("pipeline_batch", batch)
            gi, gd = index.search(hq[:Q], 5, 10)
            if Q not in ref: ref[Q] = (gi.copy(), gd.copy())
            same = np.array_equal(gi, ref[Q][0]) and np.array_equal(gd.view(np.uint32), ref[Q][1].view(np.uint32))
            reps = 12
            t0 = time.perf_counter()
            for _ in range(reps):
                index.search(hq[:Q], 5, 10)
            dt = (time.perf_counter() - t0) / reps
            print(f"Q={Q} lanes={lanes} batch={batch}: {dt * 1e3:.3f} ms  {Q / dt / 1e6:.2f} M q/s same={same}", flush=True)
