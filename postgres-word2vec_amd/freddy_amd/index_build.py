"""Offline index creation: synthetic GoogleNews-shaped corpus, k-means codebooks, encoding.

Counterpart of the reference's index_creation/ scripts (vec2database.py, pq_index.py,
ivfadc.py, ivpq.py, quantizer_creation.py), which need faiss + scipy + a live Postgres.
Index creation is NOT on the parity-checked hot path: the tables produced here are
*inputs* of the search (oracle and HIP path read the same arrays).  torch is used as
plumbing (matmul k-means on whatever device is available).
"""
import numpy as np
import torch


def _gen(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def make_corpus(N, d=300, seed=20260101, n_clusters=None, latent=30, spread=0.35, noise=0.01,
                dup_frac=0.001, device="cpu", chunk=1 << 18):
    """N x d float32, L2-normalised rows (the reference searches google_vecs_norm).

    Mixture of Gaussian clusters in a `latent`-dim subspace lifted to d dims plus a little
    isotropic noise, then normalised -- clustered like word embeddings, so PQ / IVFADC
    reach a useful recall.  `dup_frac` of the rows are exact copies of other rows (ties).
    Ids are 1..N in generation order (vec2database.py: serial ids in file order).
    """
    device = torch.device(device)
    g = _gen(seed, device)
    if n_clusters is None:
        n_clusters = max(16, min(10000, N // 300))
    centers = torch.randn(n_clusters, latent, generator=g, device=device)
    lift = torch.randn(latent, d, generator=g, device=device) / np.sqrt(latent)
    x = torch.empty(N, d, dtype=torch.float32, device=device)
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        cid = torch.randint(0, n_clusters, (e - s,), generator=g, device=device)
        z = centers[cid] + spread * torch.randn(e - s, latent, generator=g, device=device)
        v = z @ lift + noise * torch.randn(e - s, d, generator=g, device=device)
        x[s:e] = v / v.norm(dim=1, keepdim=True).clamp_min(1e-12)
    n_dup = int(N * dup_frac)
    if n_dup > 0 and N > 1:
        dst = torch.randint(0, N, (n_dup,), generator=g, device=device)
        src = torch.randint(0, N, (n_dup,), generator=g, device=device)
        x[dst] = x[src].clone()
    return x


def _assign(x, cent, chunk=1 << 16):
    """nearest centroid per row (squared L2 via the matmul expansion; index creation only)."""
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    c2 = (cent * cent).sum(1)
    for s in range(0, x.shape[0], chunk):
        xs = x[s:s + chunk]
        dist = c2[None, :] - 2.0 * (xs @ cent.T)
        out[s:s + chunk] = dist.argmin(1)
    return out


def kmeans(x, k, iters=10, seed=0):
    """Lloyd's algorithm (reference: scipy.cluster.vq.kmeans, quantizer_creation.py:13-52)."""
    n = x.shape[0]
    g = _gen(seed, x.device)
    k = int(k)
    if n >= k:
        perm = torch.randperm(n, generator=g, device=x.device)[:k]
    else:
        perm = torch.randint(0, n, (k,), generator=g, device=x.device)
    cent = x[perm].clone()
    for _ in range(iters):
        a = _assign(x, cent)
        sums = torch.zeros_like(cent).index_add_(0, a, x)
        cnt = torch.bincount(a, minlength=k).to(x.dtype)
        empty = cnt == 0
        cent = sums / cnt.clamp_min(1.0)[:, None]
        if empty.any():   # re-seed empty cells from random points
            ne = int(empty.sum())
            cent[empty] = x[torch.randint(0, n, (ne,), generator=g, device=x.device)]
    return cent


def train_pq(x, m, K, iters=10, seed=0):
    """m sub-codebooks of K centroids over d/m-dim slices -> [m][K][d/m]."""
    d = x.shape[1]
    s = d // m
    return torch.stack([kmeans(x[:, p * s:(p + 1) * s].contiguous(), K, iters, seed + 17 * p) for p in range(m)])


def encode_pq(x, codebook, chunk=1 << 16):
    """exact 1-NN code per sub-vector (reference: faiss IndexFlatL2, pq_index.py:31-63) -> int16 [N][m]."""
    m, K, s = codebook.shape
    codes = torch.empty(x.shape[0], m, dtype=torch.int16, device=x.device)
    for p in range(m):
        codes[:, p] = _assign(x[:, p * s:(p + 1) * s].contiguous(), codebook[p], chunk).to(torch.int16)
    return codes


def _np(t):
    return t.detach().cpu().numpy()


def build_pq_index(x, m=12, K=256, train_size=100000, iters=10, seed=1):
    """pq_codebook + pq_quantization (pq_index.py)."""
    cb = train_pq(x[:train_size], m, K, iters, seed)
    codes = encode_pq(x, cb)
    return dict(codebook=_np(cb).astype(np.float32), ids=np.arange(1, x.shape[0] + 1, dtype=np.int32),
                codes=_np(codes).astype(np.int16))


def build_ivf_index(x, C=1000, m=12, K=256, train_size=100000, iters=10, seed=2):
    """coarse_quantization + residual_codebook + fine_quantization as inverted lists (ivfadc.py)."""
    N = x.shape[0]
    coarse = kmeans(x[:train_size], C, iters, seed)
    cid = _assign(x, coarse)
    cb = train_pq(x[:train_size] - coarse[cid[:train_size]], m, K, iters, seed + 1000)
    codes = torch.empty(N, m, dtype=torch.int16, device=x.device)
    chunk = 1 << 18
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        codes[s:e] = encode_pq(x[s:e] - coarse[cid[s:e]], cb)
    order = torch.sort(cid, stable=True).indices      # ids stay ascending inside each list
    counts = torch.bincount(cid, minlength=C)
    list_off = np.zeros(C + 1, np.int32)
    list_off[1:] = np.cumsum(_np(counts))
    return dict(coarse=_np(coarse).astype(np.float32), codebook=_np(cb).astype(np.float32),
                list_off=list_off, ids=(_np(order) + 1).astype(np.int32),
                codes=_np(codes[order]).astype(np.int16), coarse_id=_np(cid).astype(np.int32))


def build_ivpq_index(x, m=30, K=32, k_coarse=32, train_size=100000, iters=10, seed=3, keep_vectors=True):
    """codebook_ivpq + coarse_quantization_ivpq (2-position multi index) + fine_quantization_ivpq
    + statistics over the whole corpus (ivpq.py; create_statistics, freddy--0.0.1.sql:150-186)."""
    N, d = x.shape
    half = d // 2
    cq = torch.stack([kmeans(x[:train_size, p * half:(p + 1) * half].contiguous(), k_coarse, iters, seed + p)
                      for p in range(2)])
    c0 = _assign(x[:, :half].contiguous(), cq[0])
    c1 = _assign(x[:, half:2 * half].contiguous(), cq[1])
    cell = (c0 + k_coarse * c1).to(torch.int32)
    cb = train_pq(x[:train_size], m, K, iters, seed + 100)
    codes = encode_pq(x, cb)
    cells = k_coarse * k_coarse
    cnt = torch.bincount(cell.to(torch.int64), minlength=cells).to(torch.float32)
    stats = np.zeros(cells + 1, np.float32)
    stats[:cells] = _np(cnt) / np.float32(N)
    stats[cells] = np.float32(N)
    return dict(codebook=_np(cb).astype(np.float32), coarse=_np(cq).astype(np.float32),
                ids=np.arange(1, N + 1, dtype=np.int32), coarse_id=_np(cell).astype(np.int32),
                codes=_np(codes).astype(np.int16),
                vectors=_np(x).astype(np.float32) if keep_vectors else None, stats=stats)


def exact_topk(x, queries, k, chunk=1 << 18):
    """Ground truth for recall: exact top-k by squared L2 over all rows (ties -> lowest id). ids are 1-based."""
    best_d = torch.full((queries.shape[0], k), float("inf"), device=x.device)
    best_i = torch.zeros((queries.shape[0], k), dtype=torch.int64, device=x.device)
    q2 = (queries * queries).sum(1)
    for s in range(0, x.shape[0], chunk):
        xs = x[s:s + chunk]
        dist = q2[:, None] + (xs * xs).sum(1)[None, :] - 2.0 * (queries @ xs.T)
        dd, ii = torch.topk(dist, min(k, xs.shape[0]), dim=1, largest=False)
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + s + 1], 1)
        sel = torch.topk(cat_d, k, dim=1, largest=False).indices
        best_d = torch.gather(cat_d, 1, sel)
        best_i = torch.gather(cat_i, 1, sel)
    return _np(best_i).astype(np.int32)


def recall_at_k(approx_ids, exact_ids):
    """The reference's "precision": |approx top-k intersect exact top-k| / k, averaged (evaluation_utils.py:230-238)."""
    hits = 0
    for a, e in zip(approx_ids, exact_ids):
        hits += len(set(int(v) for v in a if v >= 0) & set(int(v) for v in e))
    return hits / float(exact_ids.size)


def build_ivf_index_native(x, C=1000, m=12, K=256, train_size=100000, iters=10, seed=2, device=0):
    """ivfadc.py on the native index-build ABI only (no torch): quantizer training = freddy_gpu_kmeans
    (quantizer_creation.py:13-52), coarse assignment + residual codes = freddy_gpu_encode (ivfadc.py:36-96).
    x: numpy [N][d] float32.  Same dictionary as build_ivf_index."""
    from . import gpu
    x = np.ascontiguousarray(x, dtype=np.float32)
    N = x.shape[0]
    train = x[:train_size]
    rng = np.random.default_rng(seed)
    coarse, a = gpu.kmeans(train, C, iters, rng.choice(len(train), C, replace=len(train) < C).astype(np.int32), device)
    cb = gpu.train_pq_codebook(train - coarse[a], m, K, iters, seed + 1000, device)
    cid, codes = gpu.encode(cb, x, coarse=coarse, device=device)
    order = np.argsort(cid, kind="stable")            # ids stay ascending inside each list
    list_off = np.zeros(C + 1, np.int32)
    list_off[1:] = np.cumsum(np.bincount(cid, minlength=C))
    return dict(coarse=coarse, codebook=cb, list_off=list_off, ids=(order + 1).astype(np.int32),
                codes=np.ascontiguousarray(codes[order]), coarse_id=cid.astype(np.int32))
