#!/usr/bin/env python3
"""Per-phase shader-clock sums of the scan kernel (ivf_filter5_kernel's PROF instantiation) on the bench workload.
Needs the lab build:  FREDDY_BUILD_TAG=lab FREDDY_HIPCC_EXTRA=-DFREDDY_LAB python -c 'import __graft_entry__ as g; g.build()'
then on the GPU box:  FREDDY_GPU_SO=postgres-word2vec_amd/libfreddy_gpu_lab.so python tools/lab/scan_prof.py [N] [Q]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
import numpy as np, torch
from freddy_amd import gpu, index_build as ib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=int(os.environ.get("K", "1024")), train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
rng = np.random.default_rng(7)
qid = np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False))
dq = x[torch.from_numpy(qid - 1).to(dev)].contiguous()
res = torch.zeros((2, Q, 5), dtype=torch.int32, device=dev)
st = torch.zeros(4, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(dev)
def run(n):
    for _ in range(n):
        index.search_dev(dq.data_ptr(), Q, 5, 10, 1000.0, gpu.FOUND_ROWS, res[0].data_ptr(), res[1].data_ptr(), st.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(dev)
run(5)
if os.environ.get("FENCE"):
    index.set_option("scan_fence", int(os.environ["FENCE"]))   # (parts of the kernel off: tools/lab/ablate.py)
index.set_option("fused_prof", 1)
run(3)
index.set_option("fused_prof", 0)
index.profile_enable(True)
run(20)
prof = index.profile_read()
print({n: round(1e3 * ms / max(l, 1), 2) for n, (l, ms) in prof.items()})
