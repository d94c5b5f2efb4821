"""Multi-GPU query sharding: the index is replicated on every GPU, the batch of
queries is split contiguously across ranks, every rank searches its slice, and the per-shard
top-k lists are gathered (RCCL all_gather over xGMI when the backend is "nccl"; gloo on CPU
for the tests).  There is no exchange step during the search itself -- queries are independent
(freddy.c:835-982 keeps no cross-query state)."""
import torch
import torch.distributed as dist


def shard_bounds(Q, rank, world):
    """Rank r owns queries [lo, hi): contiguous, sizes differ by at most one."""
    base, rem = divmod(Q, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_topk(ids, dist_, Q, group=None):
    """ids/dist_: this rank's [q_local, k] tensors (same device).  Returns the [Q, k] results of the
    whole batch on every rank.  Shards may differ by one row, so they are padded to the largest
    shard for the fixed-size all_gather and trimmed afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return ids, dist_
    rank = dist.get_rank(group)
    k = ids.shape[1]
    per = (Q + world - 1) // world
    pad_i = torch.full((per, k), -1, dtype=ids.dtype, device=ids.device)
    pad_d = torch.zeros((per, k), dtype=dist_.dtype, device=dist_.device)
    lo, hi = shard_bounds(Q, rank, world)
    pad_i[:hi - lo] = ids
    pad_d[:hi - lo] = dist_
    all_i = torch.empty((world * per, k), dtype=ids.dtype, device=ids.device)
    all_d = torch.empty((world * per, k), dtype=dist_.dtype, device=dist_.device)
    dist.all_gather_into_tensor(all_i, pad_i, group=group)
    dist.all_gather_into_tensor(all_d, pad_d, group=group)
    out_i, out_d = [], []
    for r in range(world):
        a, b = shard_bounds(Q, r, world)
        out_i.append(all_i[r * per:r * per + (b - a)])
        out_d.append(all_d[r * per:r * per + (b - a)])
    return torch.cat(out_i), torch.cat(out_d)


def sharded_search(search_fn, queries, k, group=None):
    """search_fn(local_queries) -> (ids[q,k], dist[q,k]) torch tensors; returns the full batch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    Q = queries.shape[0]
    lo, hi = shard_bounds(Q, rank, world)
    ids, dd = search_fn(queries[lo:hi])
    return gather_topk(ids, dd, Q, group)


class PipelinedGather:
    """Double-buffered, asynchronous gather of the per-shard top-k (what bench.py times).

    ids and distances of a shard live in ONE [2][q_local][k] int32 buffer (row 0 = ids, row 1 = the
    distances' bits), so a step's results cross xGMI in a single all_gather (40 KB per rank at Q=1024, k=5:
    pure latency).  `depth` such buffers alternate: the gather of step i only has to be finished before its
    buffers are reused by step i+depth, so its latency hides under the next step's kernels.

    Ordering contract: the search of a step must be enqueued on torch's CURRENT stream (pass its handle to
    the C ABI) between next_buffer() and submit().  The collective is enqueued behind that stream's work,
    and wait() makes the current stream wait for it before the buffer is written again."""

    def __init__(self, q_local, k, device, group=None, depth=2):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group, self.depth = group, depth
        self.res = [torch.zeros((2, q_local, k), dtype=torch.int32, device=device) for _ in range(depth)]
        self.gathered = ([torch.zeros((self.world, 2, q_local, k), dtype=torch.int32, device=device) for _ in range(depth)]
                         if self.world > 1 else None)
        self.pending = [None] * depth
        self.steps = 0
        self.cur = 0

    def next_buffer(self):
        """The [2][q_local][k] buffer of the coming step (its previous gather is waited for first)."""
        b = self.steps % self.depth
        self.steps += 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        self.cur = b
        return self.res[b]

    def submit(self):
        """Start the gather of the buffer handed out last."""
        if self.world > 1:
            b = self.cur
            self.pending[b] = dist.all_gather_into_tensor(self.gathered[b].view(-1), self.res[b].view(-1),
                                                          group=self.group, async_op=True)

    def drain(self):
        for b in range(self.depth):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def last(self):
        """(local buffer, gathered [world][2][q_local][k] or None) of the most recent step; drain() first."""
        return self.res[self.cur], (self.gathered[self.cur] if self.gathered is not None else None)
