"""GPU-box helper: does the host-buffer pipeline keep its overlap when the process owns other streams (bench.py creates
four torch streams before the first host-buffer call)?   python tools/pipe_queues.py <n_torch_streams_before>"""
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
n_pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
streams = [torch.cuda.Stream(dev) for _ in range(n_pre)]
for st in streams:
    with torch.cuda.stream(st):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
rng = np.random.default_rng(7)
qid = rng.choice(np.arange(1, N + 1), size=8192, replace=False)
hq = x[torch.from_numpy(qid - 1).to(dev)].cpu().numpy()
for Q in (1024, 4096, 8192):
    index.search(hq[:Q], 5, 10)
    t0 = time.perf_counter()
    for _ in range(8):
        index.search(hq[:Q], 5, 10)
    dt = (time.perf_counter() - t0) / 8
    print(f"{n_pre} torch streams first, GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}: Q={Q}: {dt * 1e3:.3f} ms  {Q / dt / 1e6:.2f} M q/s", flush=True)
