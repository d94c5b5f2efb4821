"""GPU-box helper: is the bench step CPU-bound?  Time to ENQUEUE n steps (no sync) vs time until they are done."""
import os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "postgres-word2vec_amd")]
import numpy as np, torch
from freddy_amd import gpu, index_build as ib
dev = torch.device("cuda", 0)
x = ib.make_corpus(3000000, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
rng = np.random.default_rng(7)
qids = np.sort(rng.choice(np.arange(1, 3000001), size=1024, replace=False)).astype(np.int64)
d_q = x[torch.from_numpy(qids - 1).to(dev)].contiguous()
res = torch.empty((2, 1024, 5), dtype=torch.int32, device=dev)
st = torch.zeros(4, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
def step():
    index.search_dev(d_q.data_ptr(), 1024, 5, 10, 1000.0, gpu.FOUND_ROWS, res[0].data_ptr(), res[1].data_ptr(), st.data_ptr(), stream.cuda_stream)
for _ in range(20): step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6*(t1-t0)/n:.1f} us per step, done after {1e6*(t2-t0)/n:.1f} us per step")
