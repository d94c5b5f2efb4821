"""GPU-box helper: host timeline (option pipe_trace) + per-kernel durations of ONE synchronous host-buffer call of Q queries."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import gpu, index_build as ib
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
N = 3_000_000
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"])
rng = np.random.default_rng(7)
qid = rng.choice(np.arange(1, N + 1), size=Q, replace=False)
hq = x[torch.from_numpy(qid - 1).to(dev)].cpu().numpy()
for _ in range(5):
    index.search(hq, 5, 10)
reps = 30
t0 = time.perf_counter()
for _ in range(reps):
    index.search(hq, 5, 10)
print(f"Q={Q}: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms per call (pageable)")
pb = gpu.PinnedBuffer((Q, 300)); pb.array[:] = hq
for _ in range(3):
    index.search(pb.array, 5, 10)
t0 = time.perf_counter()
for _ in range(reps):
    index.search(pb.array, 5, 10)
print(f"Q={Q}: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms per call (pinned query buffer: no staging copy)")
index.profile_enable(True)
for _ in range(10):
    index.search(hq, 5, 10)
prof = index.profile_read()
index.profile_enable(False)
print({n: round(1e3 * ms / max(l, 1), 2) for n, (l, ms) in prof.items()}, "sum", round(sum(1e3 * ms / max(l, 1) for l, ms in prof.values()), 1))
os.environ["FREDDY_GPU_PIPE_TRACE"] = "1"   # (read by -DFREDDY_LAB builds: FREDDY_GPU_SO=...libfreddy_gpu_lab.so)
for _ in range(3):
    index.search(hq, 5, 10)
t0 = time.perf_counter(); hq2 = hq.copy(); print(f"host memcpy of the queries ({hq.nbytes} B): {(time.perf_counter() - t0) * 1e6:.0f} us")
