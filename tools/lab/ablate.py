#!/usr/bin/env python3
"""What the scan kernel's parts cost: ivf_filter5_kernel with parts switched OFF through FilterArgs::fence (lab builds only; the
results are wrong, only the time means something).  bits: 2 gathers, 4 selection tail, 16 slab stores, 32 table loads,
64 code loads of phases >= 2, 128 the slow survivor path.
  FREDDY_BUILD_TAG=lab FREDDY_HIPCC_EXTRA=-DFREDDY_LAB python -c 'import __graft_entry__ as g; g.build()'
  FREDDY_GPU_SO=postgres-word2vec_amd/libfreddy_gpu_lab.so python tools/lab/ablate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
import numpy as np, torch
from freddy_amd import gpu, index_build as ib

N, Q = 3_000_000, 1024
dev = torch.device("cuda", 0)
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=int(os.environ.get("K", "1024")), train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
if os.environ.get("K", "1024") != "1024":
    index.set_option("codes_u8", 2)   # (the six-phase kernel's one-byte instantiation: fused8.h has no switches)
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
cases = [("everything on", 0), ("no selection tail", 4), ("no gathers", 2), ("no table loads", 32), ("no slab stores", 16),
         ("no table loads, no slab stores", 48), ("no later code loads", 64), ("no gathers, no tail", 6),
         ("no loads, stores, gathers", 2 | 16 | 32 | 64), ("skeleton (barriers, records, row terms, first code words)", 2 | 4 | 16 | 32 | 64)]
for rnd in range(2):
    for name, f in cases:
        index.set_option("scan_fence", f)
        run(5)
        index.profile_enable(True)
        run(40)
        prof = index.profile_read()
        index.profile_enable(False)
        l, ms = prof["ivf_filter"]
        print(f"round {rnd}  fence {f:3d}  {name:60s} {1e3 * ms / l:7.2f} us", flush=True)
index.set_option("scan_fence", 0)
