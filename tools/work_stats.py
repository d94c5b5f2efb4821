"""GPU-box helper: shape of the fused kernel's work table for the bench workload (how many
(group, chunk) entries, items per entry, rows per entry) -- computed with torch from the same
corpus/index/queries bench.py uses."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]
from freddy_amd import index_build as ib

dev = torch.device("cuda", 0)
N, C, Q, W, G, CH = 3_000_000, 1000, 1024, 10, 16, 4096
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=C, m=12, K=1024, train_size=100000, iters=10, seed=2)
rng = np.random.default_rng(7)
qids = np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False)).astype(np.int64)
q = x[torch.from_numpy(qids - 1).to(dev)]
coarse = torch.from_numpy(tab["coarse"]).to(dev)
dist = torch.cdist(q, coarse)
cells = dist.topk(W, largest=False).indices.cpu().numpy()
cnt = np.bincount(cells.ravel(), minlength=C)
lens = np.diff(tab["list_off"])
entries = []
for c in range(C):
    n = cnt[c]
    chunks = -(-lens[c] // CH)
    f = 0
    while f < n:
        g = min(G, n - f)
        for ch in range(chunks):
            rows = min(CH, lens[c] - ch * CH)
            entries.append((g, rows))
        f += G
e = np.array(entries)
out = {"entries": len(e), "per_cu": len(e) / 256, "items_hist": np.bincount(e[:, 0], minlength=17).tolist(),
       "mean_items": float(e[:, 0].mean()), "mean_rows": float(e[:, 1].mean()),
       "rows_pct": np.percentile(e[:, 1], [5, 25, 50, 75, 95]).tolist(),
       "cells_touched": int((cnt > 0).sum()), "max_cnt": int(cnt.max()),
       "build_units(items)": int(e[:, 0].sum()), "build_units_padded_even": int(((e[:, 0] + 1) // 2 * 2).sum()),
       "gather_units(rows*items)": int((e[:, 0] * e[:, 1]).sum()), "gather_units_always16": int((16 * ((e[:, 1] + 511) // 512 * 512)).sum()),
       "list_len_pct": np.percentile(lens, [0, 5, 50, 95, 100]).tolist()}
print(json.dumps(out, indent=1))
