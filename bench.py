#!/usr/bin/env python3
"""bench.py -- batched IVFADC kNN throughput (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic queries that are already
resident in HBM: coarse distances (+ the query x codebook table on a side stream) -> probe plan (items
bucketed by cell) -> work table -> entry records -> the filter kernel (bounded cheap distances from two
streamed tables, LDS slabs, sums and survivor selection) -> merge with the exact stage
(the reference's arithmetic for the rows that can matter) and the updateTopK replay
(+ the asynchronous RCCL gather of the per-shard top-k when N > 1).  Results are checked bit for bit
against the CPU oracle on the bench queries (cpu_baseline.parity_with_gpu_on_sample).

Workload (BASELINE.json configs[2]): 3,000,000 x 300-d synthetic GoogleNews-shaped corpus,
C=1000 coarse cells, m=12, K=1024 residual PQ, nprobe W=10, k=5, 1024 queries per GPU
(replicated index, queries sharded by rank -> "weak" scaling).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     dominant kernel, algorithmic bytes / its HIP-event duration vs 8 TB/s HBM
  "cpu_baseline": the CPU oracle (oracle/, a port of the reference loops) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--N", type=int, default=3_000_000)
    ap.add_argument("--Q", type=int, default=1024, help="queries per GPU per step")
    ap.add_argument("--C", type=int, default=1000)
    ap.add_argument("--m", type=int, default=12)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--nprobe", type=int, default=10)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=1024, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()   # one rank per GPU; wraps only in single-GPU dry runs
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    from freddy_amd import gpu, index_build as ib

    def log(*x):
        if rank == 0:
            print("[bench]", *x, file=sys.stderr, flush=True)

    # ---- synthetic corpus + index (every rank builds the identical replica) -----------------
    t0 = time.time()
    x = ib.make_corpus(a.N, d=300, seed=20260101, device=dev)
    tab = ib.build_ivf_index(x, C=a.C, m=a.m, K=a.K, train_size=100000, iters=10, seed=2)
    log(f"corpus+index built in {time.time() - t0:.1f}s")
    t0 = time.time()
    index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=dev_index)
    log(f"pinned {index.nbytes / 1e6:.1f} MB in {time.time() - t0:.1f}s")

    # queries = indexed vectors themselves (ivfadc_batch_search takes ids), distinct per rank
    rng = np.random.default_rng(7 + rank)
    qids = np.sort(rng.choice(np.arange(1, a.N + 1), size=a.Q, replace=False)).astype(np.int64)
    d_q = x[torch.from_numpy(qids - 1).to(dev)].contiguous()
    # ids and distances of the shard live in ONE buffer so that the per-shard top-k crosses xGMI
    # in a single RCCL all_gather (the payload is 40 KB per rank: pure latency).  Two such buffers
    # alternate: the gather of step i is asynchronous and only has to be finished before its buffers are
    # reused by step i+2, so its latency hides under the next step's kernels instead of adding to them.
    d_res2 = [torch.empty((2, a.Q, a.k), dtype=torch.int32, device=dev) for _ in range(2)]
    d_res = d_res2[0]
    d_ids = d_res[0]
    d_dist = d_res[1].view(torch.float32)
    d_status = torch.zeros(4, dtype=torch.int32, device=dev)
    g_res2 = [torch.empty((world, 2, a.Q, a.k), dtype=torch.int32, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    step_no = [0]

    stream = torch.cuda.current_stream(dev)

    def step():
        b = step_no[0] & 1
        step_no[0] += 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        res = d_res2[b]
        index.search_dev(d_q.data_ptr(), a.Q, a.k, a.nprobe, 1000.0, gpu.FOUND_ROWS, res[0].data_ptr(),
                         res[1].data_ptr(), d_status.data_ptr(), stream.cuda_stream)
        if world > 1:
            pending[b] = dist.all_gather_into_tensor(g_res2[b].view(-1), res.view(-1), async_op=True)

    def barrier():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    qps = world * a.Q * a.steps / dt

    # ---- per-kernel durations with HIP events on the launch stream (instrumented re-run) ----
    index.profile_enable(True)
    for _ in range(a.steps):
        step()
    barrier()
    prof = index.profile_read()
    index.profile_enable(False)
    step_no[0] = 0
    step()      # leave this rank's final results in d_res2[0] (= d_ids / d_dist below)
    barrier()
    scanned_rows = index.last_scanned_rows()
    bound_violations = index.bound_violations()   # self-check of the filter + refine scan (must be 0)
    straggler = int(d_status[0].item())

    # ---- the same batch through the synchronous host-buffer ABI (H2D of queries, D2H of results,
    # one stream sync per call): reported beside `value`, never as `value`
    host_qps = None
    if rank == 0:
        h_q = d_q.cpu().numpy()
        index.search(h_q, a.k, a.nprobe)
        t0 = time.perf_counter()
        for _ in range(10):
            index.search(h_q, a.k, a.nprobe)
        host_qps = 10 * a.Q / (time.perf_counter() - t0)

    out = None
    if rank == 0:
        # algorithmic bytes per query: sum of probed list lengths * (m*2 + 4) + query + result
        bytes_per_launch = scanned_rows * (a.m * 2 + 4) + a.Q * (300 * 4 + a.k * 8)
        kern = {n: {"launches": l, "avg_us": round(1e3 * ms / max(l, 1), 2)} for n, (l, ms) in prof.items()}
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0] if prof else None
        roof = None
        if dom:
            avg_s = prof[dom][1] / max(prof[dom][0], 1) / 1e3
            ach = bytes_per_launch / avg_s / 1e9
            # HBM bytes of the dominant kernel from the committed rocprofv3 PMC passes of this same
            # command (profiles/latest_pmc.json, written by tools/profile_round.sh): 2 x FETCH_SIZE
            # (gfx950 reports half of wide coalesced reads) + WRITE_SIZE, KiB -> bytes.
            traffic = None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
                kname = {"ivf_fused": "ivf_filter_kernel", "adc_scan": "adc_scan_kernel", "lut_build": "lut_build_kernel",
                         "coarse_dist": "coarse_tile_kernel", "probe_plan": "probe_plan_kernel"}.get(dom, dom)
                if kname not in pmc and dom == "ivf_fused":   # FREDDY_GPU_FUSED_KERNEL=2 / 1
                    kname = "ivf_spec_kernel" if "ivf_spec_kernel" in pmc else "ivf_fused_kernel"
                if kname in pmc:
                    traffic = int((2 * pmc[kname].get("fetch_kib", 0) + pmc[kname].get("write_kib", 0)) * 1024)
            except Exception:
                traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(bytes_per_launch),
                    "avg_launch_us": round(avg_s * 1e6, 2),
                    "note": "achieved = SURVEY 8d algorithmic bytes (every probed row's 28 B once per QUERY) / the scan "
                            "kernel's duration; the kernel reads a list once per CELL entry (traffic = 2*FETCH_SIZE + "
                            "WRITE_SIZE of the committed PMC passes), so achieved can exceed what crosses HBM"}

        # ---- recall@5 vs exact search, and parity of a sample against the oracle ---------------
        recall = None
        if not a.no_recall:
            exact = ib.exact_topk(x, d_q, a.k)
            recall = ib.recall_at_k(d_ids.cpu().numpy(), exact)

        cpu = None
        if a.cpu_sample > 0 and world == 1:   # reported at N=1 only
            from oracle.oracle import Oracle
            o = Oracle()
            ot = o.ivf_table(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"])
            ns = min(a.cpu_sample, a.Q)
            cores = os.cpu_count() or 1
            qs = d_q[:ns].cpu().numpy()
            t0 = time.perf_counter()
            exp = o.ivfadc_search_many(ot, qs, a.k, a.nprobe, sentinel=1000.0, found_rule=0, n_threads=cores)
            cdt = time.perf_counter() - t0
            got_i = d_ids[:ns].cpu().numpy()
            got_d = d_dist[:ns].cpu().numpy()
            parity = bool(np.array_equal(exp["id"], got_i) and
                          np.array_equal(exp["dist"].view(np.uint32), got_d.view(np.uint32)))
            n1 = min(64, ns)   # one thread = one PostgreSQL backend
            t0 = time.perf_counter()
            o.ivfadc_search_many(ot, qs[:n1], a.k, a.nprobe, sentinel=1000.0, found_rule=0, n_threads=1)
            one_core = n1 / (time.perf_counter() - t0)
            cpu = {"value": round(ns / cdt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                   "value_1_core": round(one_core, 2),
                   "sample": f"first {ns} of the {a.Q} bench queries, same index, nprobe={a.nprobe}, k={a.k}; "
                             f"oracle/ (C port of freddy.c:174-393 loops, gcc -O2, OpenMP over queries)",
                   "parity_with_gpu_on_sample": parity}

        out = {
            "metric": "batched IVFADC kNN queries/sec (k=5) + recall@5, 3Mx300d",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "IVFADC batch (ivfadc_batch_search generalised to nprobe), "
                                   f"{a.N}x300d, C={a.C}, m={a.m}, K={a.K}, nprobe={a.nprobe}, k={a.k}, "
                                   f"batch={a.Q} queries per GPU, replicated index, queries sharded by rank",
                       "N": a.N, "d": 300, "C": a.C, "m": a.m, "K": a.K, "nprobe": a.nprobe, "k": a.k,
                       "batch_per_gpu": a.Q, "parallelism": f"dp{world}"},
            "recall_at_5": None if recall is None else round(recall, 4),
            "queries_needing_extra_round": straggler,
            "filter_bound_violations": bound_violations,
            "host_buffer_abi_queries_per_s": None if host_qps is None else round(host_qps, 1),
            "roofline": roof, "kernels": kern, "cpu_baseline": cpu,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
