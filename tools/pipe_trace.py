"""GPU-box helper: host-buffer pipeline of freddy_gpu_ivfadc_search on the bench index -- throughput against the
sub-batch size / number of lanes, and (option pipe_trace) the host timeline of one call.
  python tools/pipe_trace.py [--trace]"""
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
qid = rng.choice(np.arange(1, N + 1), size=8192, replace=False)
hq = x[torch.from_numpy(qid - 1).to(dev)].cpu().numpy()
for Q in (1024, 4096, 8192):
    for lanes in (1, 2, 4):
        for batch in (1024, 2048):
            index.set_option("pipeline_lanes", lanes); index.set_option("pipeline_batch", batch)
            index.search(hq[:Q], 5, 10)
            t0 = time.perf_counter()
            for _ in range(5):
                index.search(hq[:Q], 5, 10)
            dt = (time.perf_counter() - t0) / 5
            print(f"Q={Q} lanes={lanes} batch={batch}: {dt * 1e3:.3f} ms  {Q / dt / 1e6:.2f} M q/s", flush=True)
if "--trace" in sys.argv:
    index.set_option("pipeline_lanes", 4); index.set_option("pipeline_batch", 1024)   # (the host timeline: FREDDY_GPU_PIPE_TRACE=1 with a -DFREDDY_LAB build)
    index.search(hq[:4096], 5, 10)
    index.search(hq[:8192], 5, 10)
