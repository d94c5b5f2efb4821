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
qid = rng.choice(np.arange(1, N + 1), size=16384, replace=False)
hq = x[torch.from_numpy(qid - 1).to(dev)].cpu().numpy()
for Q in (2048, 3000, 4096, 8192, 16384):
    for lanes in (2, 3, 4):
        for batch in (1024, 1536, 2048, 3072, 4096):
            if batch * 1 > Q and batch != 1024: continue
            index.set_option("pipeline_lanes", lanes); index.set_option("pipeline_batch", batch)
            index.search(hq[:Q], 5, 10)
            t0 = time.perf_counter()
            for _ in range(6):
                index.search(hq[:Q], 5, 10)
            dt = (time.perf_counter() - t0) / 6
            print(f"Q={Q} lanes={lanes} batch={batch}: {dt * 1e3:.3f} ms  {Q / dt / 1e6:.2f} M q/s", flush=True)
