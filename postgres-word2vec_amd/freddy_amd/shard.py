"""Multi-GPU query sharding: the index is replicated on every GPU, the batch of
queries is split contiguously across ranks, every rank searches its slice, and the per-shard
top-k lists are gathered (RCCL all_gather over xGMI when the backend is "nccl"; gloo on CPU
for the tests).  There is no exchange step during the search itself -- queries are independent
(freddy.c:835-982 keeps no cross-query state)."""
import os

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


class _AfterEvent:
    """What next_buffer() waits for when the collective ran on the step's OWN stream: an event behind it."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)


class PipelinedGather:
    """Pipelined, asynchronous gather of the per-shard top-k (what bench.py times).

    ids and distances of a shard live in ONE [2][q_local][k] int32 buffer (row 0 = ids, row 1 = the
    distances' bits), so a step's results cross xGMI in a single all_gather (40 KB per rank at Q=1024, k=5:
    pure latency).  `depth` such buffers alternate: the gather of step i only has to be finished before its
    buffer is reused by step i+depth, so its latency hides under the next steps' kernels.

    gather_every = G > 1: the buffers form a ring of depth / G GROUPS of G consecutive steps, and ONE all_gather
    moves a whole group (fewer, larger collectives: with four batches in flight on four streams a collective per
    step is a fifth active stream beside them all the time -- hardware queues, bench.py -- and costs the host a
    collective call per 0.1 ms step).  The group's gather is enqueued behind the LAST of its steps after that
    step's stream has been made to wait for the other steps' events; a group is rewritten depth steps later, so a
    ring of two groups (depth = 2 G) leaves a whole group of slack.

    Ordering contract: the search of a step must be enqueued on the stream passed to next_buffer() / submit()
    (torch's CURRENT stream when none is passed) between the two calls.  The collective is enqueued behind that
    stream's work, and next_buffer() makes the stream wait for the gather that last read the buffer.

    force: run the collective path with a single rank too (a 1-rank process group: bench.py --force-collective).

    in_stream: the collective is called with async_op=False under the step's stream, which makes ProcessGroupNCCL enqueue the
    RCCL kernel ON THAT STREAM (no internal communication stream, no event hop in and out of it: the searching streams stay the
    only active ones -- with four batches in flight every further active stream costs hardware-queue sharing, bench.py); the
    host does not block (stream-ordered); later writers of the group wait for an event recorded behind the collective."""

    def __init__(self, q_local, k, device, group=None, depth=2, force=False, gather_every=1, in_stream=False, comm=None):
        # comm: a freddy_amd.rccl.Communicator, or {stream handle: Communicator} with one communicator per searching stream -- the
        # gather is ONE ncclAllGather call on the step's stream (no c10d work object, events or stream context: rccl.py has the
        # measurements); implies in_stream.  One communicator per stream: RCCL orders the operations of ONE communicator among
        # themselves (an operation issued on another stream than its predecessor first waits for it), which would tie the four
        # searching streams' chains together rank by rank; with a communicator each, stream order is the only dependency
        self.comm = comm
        self.bound = {}
        self.in_stream = bool(in_stream) or comm is not None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = (self.world > 1 or force) and dist.is_initialized()
        G = max(1, int(gather_every))
        if depth % G:
            raise ValueError("depth must be a multiple of gather_every")
        self.group, self.depth, self.G = group, depth, G
        self.cuda = torch.device(device).type == "cuda"
        self.ring = torch.zeros((depth, 2, q_local, k), dtype=torch.int32, device=device)
        self.res = [self.ring[b] for b in range(depth)]
        self.n_groups = depth // G
        self.gathered = ([torch.zeros((self.world, G, 2, q_local, k), dtype=torch.int32, device=device) for _ in range(self.n_groups)]
                         if self.collective else None)
        self.pending = [None] * self.n_groups
        self.events = [torch.cuda.Event() for _ in range(depth)] if (self.cuda and G > 1) else None
        self.steps = 0
        self.cur = 0
        self.unsent = 0      # steps of the current group whose gather has not been enqueued yet

    def _ctx(self, stream):
        import contextlib
        return torch.cuda.stream(stream) if (stream is not None and self.cuda) else contextlib.nullcontext()

    def next_buffer(self, stream=None):
        """The [2][q_local][k] buffer of the coming step (the gather that last read it is waited for first)."""
        b = self.steps % self.depth
        self.steps += 1
        g = b // self.G
        if self.pending[g] is not None:
            with self._ctx(stream):
                self.pending[g].wait()          # (the given stream waits; the host does not unless the backend is gloo)
            if b % self.G == self.G - 1:        # every stream that writes into the group has been told
                self.pending[g] = None
        self.cur = b
        return self.res[b]

    def _gather_group(self, g, stream, upto):
        """all_gather of group g behind `stream`, which first waits for the group's other steps (events)."""
        with self._ctx(stream):
            if self.events is not None:
                cs = torch.cuda.current_stream()
                for j in range(upto):
                    cs.wait_event(self.events[g * self.G + j])
            lo = g * self.G
            self.pending[g] = self._all_gather(self.gathered[g].view(-1), self.ring[lo:lo + self.G].view(-1))
        self.unsent = 0

    def _all_gather(self, dst, src):
        if self.comm is not None and self.cuda:   # (grouped gathers: the caller has set the stream context and waited for the group's events)
            st = torch.cuda.current_stream()
            cm = (self.comm.get(st.cuda_stream) or next(iter(self.comm.values()))) if isinstance(self.comm, dict) else self.comm
            cm.all_gather_i32(dst, src, st)
            ev = torch.cuda.Event()
            ev.record(st)
            return _AfterEvent(ev)
        fake = os.environ.get("FREDDY_LAB_GATHER_FAKE")   # lab: "copy" = a plain device copy, "none" = nothing, in place of the collective
        if fake and self.cuda:
            if fake == "copy":
                dst[:src.numel()].copy_(src, non_blocking=True)
            return _AfterEvent(None)
        if not (self.in_stream and self.cuda):
            return dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)
        dist.all_gather_into_tensor(dst, src, group=self.group, async_op=False)   # on the CURRENT stream (set by the caller)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return _AfterEvent(ev)

    def submit(self, stream=None):
        """The step that wrote the buffer handed out last is enqueued: start the gather it completes."""
        if not self.collective:
            return
        b = self.cur
        if self.G == 1 and self.comm is not None and stream is not None:
            # RCCL directly on the step's stream.  Nothing to wait for later either: buffer b is rewritten `depth` steps on, and
            # with depth a multiple of the streams in turn that is the SAME stream -- stream order is the dependency
            fn = self.bound.get((b, stream.cuda_stream))
            if fn is None:
                cm = self.comm.get(stream.cuda_stream) if isinstance(self.comm, dict) else self.comm
                fn = self.bound[(b, stream.cuda_stream)] = cm.bind_all_gather_i32(self.gathered[b].view(-1), self.res[b].view(-1), stream)
            fn()
            return
        if self.G == 1:
            with self._ctx(stream):
                self.pending[b] = self._all_gather(self.gathered[b].view(-1), self.res[b].view(-1))
            return
        self.unsent += 1
        if b % self.G == self.G - 1:
            self._gather_group(b // self.G, stream, self.G - 1)
        elif self.events is not None:
            self.events[b].record(stream if stream is not None else torch.cuda.current_stream())

    def drain(self):
        """Every gather enqueued and waited for (a group that is only partly written is gathered as it is)."""
        if self.collective and self.G > 1 and self.unsent:
            b = self.cur          # (not the last step of its group: every step written so far has recorded its event in submit())
            self._gather_group(b // self.G, None, (b % self.G) + 1)
        for g in range(self.n_groups):
            if self.pending[g] is not None:
                self.pending[g].wait()
                self.pending[g] = None

    def last(self):
        """(local buffer, gathered [world][2][q_local][k] or None) of the most recent step; drain() first."""
        if self.gathered is None:
            return self.res[self.cur], None
        return self.res[self.cur], self.gathered[self.cur // self.G][:, self.cur % self.G]
