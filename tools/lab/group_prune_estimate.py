#!/usr/bin/env python3
"""VERDICT r5 task 7 asked for a group-level cell selection for indexes of thousands of cells: cluster the coarse centroids into
~sqrt(C) groups and compute only the 128-cell tiles of groups a triangle-inequality bracket cannot exclude.  This estimates, on the
bench corpus' own geometry and on the CPU, how many groups such a bracket excludes: centroids = k-means (13 000 cells) of the
synthetic mixture (10 000 components in a 30-dimensional latent space lifted to 300 dimensions, rows normalised), groups = k-means
of the centroids, T = a query's 10th smallest centroid distance (exact), lower bound of group g = max(0, |q - g| - radius_g)^2.
Result (round 6): 102 / 256 / 512 groups -> 100 % of the groups and cells kept: the groups' radii (0.8 - 0.9) are as large as the
distance to the W-th nearest centroid (0.90) while the group centres are 1.1 - 1.2 away -- the bracket never closes.  Not built."""
import numpy as np
import torch

torch.manual_seed(0)
g = torch.Generator().manual_seed(20260101)
ncl, latent, d, C = 10000, 30, 300, 13000
centers = torch.randn(ncl, latent, generator=g)
lift = torch.randn(latent, d, generator=g) / np.sqrt(latent)


def sample(n, spread=0.35, noise=0.01):
    cid = torch.randint(0, ncl, (n,), generator=g)
    z = centers[cid] + spread * torch.randn(n, latent, generator=g)
    v = z @ lift + noise * torch.randn(n, d, generator=g)
    return v / v.norm(dim=1, keepdim=True)


x = sample(400000)
cent = x[torch.randperm(x.shape[0], generator=g)[:C]].clone()
for _ in range(6):
    a = torch.cat([(xs @ cent.T).argmax(1) for xs in x.split(20000)])
    cent.index_add_(0, a, x)
    cent /= (torch.bincount(a, minlength=C).float() + 1)[:, None]
q = sample(500)
D = (q * q).sum(1)[:, None] + (cent * cent).sum(1)[None, :] - 2 * q @ cent.T
T = D.kthvalue(10, dim=1).values
for G in (102, 256, 512):
    gc = cent[torch.randperm(C, generator=g)[:G]].clone()
    for _ in range(8):
        a = ((cent * cent).sum(1)[:, None] + (gc * gc).sum(1)[None, :] - 2 * cent @ gc.T).argmin(1)
        gc.zero_().index_add_(0, a, cent)
        gc /= torch.bincount(a, minlength=G).float().clamp_min(1)[:, None]
    a = ((cent * cent).sum(1)[:, None] + (gc * gc).sum(1)[None, :] - 2 * cent @ gc.T).argmin(1)
    r = torch.zeros(G).scatter_reduce(0, a, (cent - gc[a]).norm(dim=1), reduce="amax")
    sizes = torch.bincount(a, minlength=G)
    dg = (q[:, None, :] - gc[None, :, :]).norm(dim=2)
    keep = (dg - r[None, :]).clamp_min(0) ** 2 <= T[:, None]
    print(f"G={G}: groups kept {keep.float().mean().item():.3f}, cells kept {(keep.float() @ sizes.float() / C).mean().item():.3f}, "
          f"mean radius {r.mean():.3f}, mean distance to a group centre {dg.mean():.3f}, mean sqrt(T) {T.sqrt().mean():.3f}")
