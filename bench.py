#!/usr/bin/env python3
"""bench.py -- batched IVFADC kNN throughput (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic queries that are already resident in
HBM: coarse distances (+ the query x codebook table on a side stream) -> probe plan (items bucketed by
cell) -> work table -> entry records -> the filter kernel (bounded cheap distances, LDS slabs, sums and
survivor selection) -> merge with the exact stage (the reference's arithmetic for the rows
that can matter) and the updateTopK replay (+ with N > 1 the RCCL gather of the per-shard top-k: one ncclAllGather
call on the step's own stream, freddy_amd/rccl.py; `--force-collective` runs that path with a one-rank group on one GPU).  Results are checked bit for bit against the CPU oracle on the bench queries
(cpu_baseline.parity_with_gpu_on_sample).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 3,000,000 x 300-d synthetic
GoogleNews-shaped corpus, C=1000 coarse cells, m=12, K=1024 residual PQ, nprobe W=10, k=5, 1024 queries per
GPU (replicated index, queries sharded by rank -> "weak" scaling; --scaling strong splits ONE batch of --Q
queries over the ranks).  --config pq / join time BASELINE configs[1] / configs[3] with the same JSON shape.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts N ranks itself (a child
`python -m torch.distributed.run`, before anything touches the GPU).  `--dry-run --backend gloo` runs the
sharded step -- double-buffered asynchronous gather included -- on CPU tensors with a stand-in search, which
is how the CPU test suite covers this file's N > 1 path.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     the dominant kernel against the HBM roofline + the ceilings that actually bind (LDS gather)
  "cpu_baseline": the CPU oracle (oracle/, a port of the reference loops) on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime starts.  The
# four batches in flight want a queue each, beside the queues the default stream, the library's own stream and -- with --gpus N --
# RCCL's stream take: wherever two of the searching streams share a queue their chains serialise (profiles/r05_queue_sweep.txt:
# 4 batches on 4 / 5 queues 5.7-5.9 M q/s, on 6 / 7 / 8 / 12 / 16 queues 9.8-9.9 M; a FIFTH active stream costs 6 queues 40 %, 8
# queues 25 %).  Eight: the same for every --gpus N, so rank 0 of an N-GPU run is the single-GPU configuration.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the CPU oracle's OpenMP workers (cpu_baseline: one per core) go to sleep after their parallel region instead of spinning on the
# cores the host side of the measurements that follow needs (the kNN-join pass right after it read 1.5 instead of 1.1 ms)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "postgres-word2vec_amd")]

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LDS_PEAK_GBS = 150000.0    # ibid. (LDS): ~150 TB/s aggregate for ds_read_b64 / b128 with every CU streaming
METRIC = "batched IVFADC kNN queries/sec (k=5) + recall@5, 3Mx300d"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="ivfadc", choices=["ivfadc", "pq", "join", "exact"],
                    help="ivfadc = BASELINE configs[2] (the metric's), pq = configs[1], join = configs[3], exact = brute-force kNN (SURVEY 8f-1)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --Q queries per GPU per step; strong: --Q queries per step in total")
    ap.add_argument("--N", type=int, default=None)
    ap.add_argument("--Q", type=int, default=None, help="queries per step (per GPU with weak scaling)")
    ap.add_argument("--C", type=int, default=1000)
    ap.add_argument("--m", type=int, default=12)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--nprobe", type=int, default=10)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=1024, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--host-abi-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--same-queries", action="store_true",
                    help="every in-flight stream searches the SAME query set (round 2's setup; the default gives each stream its own)")
    ap.add_argument("--no-host-abi", action="store_true", help="skip the host-buffer ABI measurement (a child process); profiling runs")
    ap.add_argument("--no-collective-child", action="store_true", help="skip the one-rank collective side measurement (a child process)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="config ivfadc on one GPU also runs bounded passes of --config pq and --config join and reports them "
                         "under other_configs; this switches that off")
    ap.add_argument("--stream-skip", type=int, default=0, help="experiment: create this many unused streams first")
    ap.add_argument("--scan-share", type=int, default=0, help="experiment: option scan_share of the timed region (0 = the batches in flight)")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="batches in flight per GPU (config ivfadc): consecutive steps alternate between this many HIP streams, "
                         "so the front end of batch i+1 runs beside the merge of batch i; 1 = strictly one batch at a time")
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1 through the N > 1 code path: a ONE-rank process group of --backend, the asynchronous all_gather, option "
                         "reserve_cus, the per-step ncclAllGather on the searching streams (what rank 0 of an N-GPU run does, priced on one GPU)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="steps per all_gather of the per-shard top-k (0 or 1 = a collective per step, in the step's stream; G > 1 = one "
                         "collective per group of G steps over a ring of two groups)")
    ap.add_argument("--gather-path", default="rccl", choices=["rccl", "c10d", "c10d-async"],
                    help="rccl: ONE ncclAllGather call on the searching stream of the step (freddy_amd/rccl.py: a communicator of its own over "
                         "the ranks torch.distributed started); c10d: torch.distributed.all_gather_into_tensor with async_op=False under the "
                         "step's stream; c10d-async: async_op=True on ProcessGroupNCCL's internal stream")
    ap.add_argument("--one-comm", action="store_true", help="--gather-path rccl with ONE communicator for all searching streams (default: one each)")
    ap.add_argument("--collective-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--ab-rounds", type=int, default=2, help="--force-collective: rounds of (without, with) timed regions")
    ap.add_argument("--reserve-cus", type=int, default=2, help="option reserve_cus of the collective path (CUs every persistent scan leaves free)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only with --dry-run)")
    ap.add_argument("--dry-run", action="store_true", help="CPU tensors + stand-in search: exercises the sharded step / gather only")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """--gpus N without a torchrun environment: start N ranks as children.  This process has not touched the
    GPU (importing torch does not); it only waits and passes the children's exit code on."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def log(rank, *x):
    if rank == 0:
        print("[bench]", *x, file=sys.stderr, flush=True)


def fill_pattern(res, rank, step):
    """Stand-in search of the dry run: a (rank, step) dependent pattern in the [2][q][k] result buffer."""
    n = res[0].numel()
    base = torch.arange(n, dtype=torch.int32, device=res.device).view(res[0].shape)
    res[0].copy_(base + (rank * 1_000_003 + step * 7919))
    res[1].copy_(base * 3 + (rank * 17 + step))


def sharded_steps(step_fn, pg, steps, warmup, sync, world):
    """warmup + exactly `steps` timed steps, bracketed by drain + barrier + device sync on both sides;
    returns the MAX over ranks of the elapsed seconds."""
    import torch.distributed as dist

    def barrier():
        pg.drain()
        sync()
        if world > 1 or (pg.collective and dist.is_initialized()):
            dist.barrier()
        sync()

    for _ in range(warmup):
        step_fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dev = pg.res[0].device
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt, barrier


def verify_gather(pg, rank, world):
    """The gathered buffer of the last step must hold every rank's own result: slot `rank` equals the local
    buffer, and every slot's checksum equals the checksum its owner reports."""
    import torch.distributed as dist
    if pg.gathered is None:
        return True
    local, gathered = pg.last()
    ok = bool(torch.equal(gathered[rank], local))
    mine = local.to(torch.int64).sum().view(1)
    sums = torch.zeros(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sums, mine)
    got = gathered.to(torch.int64).sum(dim=(1, 2, 3))
    ok = ok and bool(torch.equal(got, sums))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=local.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def run_dry(a, rank, world):
    """The N > 1 path of this file on CPU tensors (gloo): sharding, the double-buffered asynchronous gather,
    the timing brackets and the JSON line -- with fill_pattern() where the HIP search would be."""
    import torch.distributed as dist
    from freddy_amd import shard
    Q = a.Q or 1024
    q_local = Q if a.scaling == "weak" else shard.shard_bounds(Q, rank, world)[1] - shard.shard_bounds(Q, rank, world)[0]
    if a.scaling == "strong" and Q % world:
        raise SystemExit("--scaling strong needs --Q divisible by the number of ranks")
    G = max(1, a.gather_every)
    pg = shard.PipelinedGather(q_local, a.k, torch.device("cpu"), depth=2 * G if G > 1 else 2, gather_every=G, force=a.force_collective)
    n = [0]

    def step():
        res = pg.next_buffer()
        fill_pattern(res, rank, n[0])
        n[0] += 1
        pg.submit()

    dt, _ = sharded_steps(step, pg, a.steps, a.warmup, lambda: None, world)
    ok = verify_gather(pg, rank, world)
    if pg.gathered is not None:   # every slot must hold ITS rank's pattern of the last step
        _, gathered = pg.last()
        for r in range(world):
            exp = torch.empty_like(gathered[r])
            fill_pattern(exp, r, n[0] - 1)
            ok = ok and bool(torch.equal(gathered[r], exp))
    total_q = q_local * world if a.scaling == "weak" else Q
    out = {"metric": METRIC, "value": round(total_q * a.steps / dt, 1), "unit": "queries/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True,
           "scaling": a.scaling, "vs_baseline": None, "dtype": "none (dry run)", "data": "synthetic",
           "config": {"workload": "DRY RUN: stand-in search on CPU tensors, gloo", "batch_per_gpu": q_local,
                      "parallelism": f"dp{world}", "world_size": (dist.get_world_size() if dist.is_initialized() else 1),
                      "backend": (dist.get_backend() if dist.is_initialized() else "none (single rank)")},
           "dry_run": True, "gather_verified": ok, "roofline": None, "cpu_baseline": None}
    out["config"]["gather_every"] = pg.G
    out["config"]["collective"] = bool(pg.collective)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(headline(out)), flush=True)
    return 0 if ok else 1


def pmc_traffic(kernel, config=None, shape=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/latest_pmc.json, or latest_pmc_<config>.json for --config pq / join; written by tools/profile_round.sh):
    2 x FETCH_SIZE (gfx950 reports half of wide coalesced reads) + WRITE_SIZE, KiB -> bytes.  The file records the
    workload shape it was taken on ("_shape"); None if there is no record or the shape differs from this run's."""
    import glob
    first = os.path.join(ROOT, "profiles", f"latest_pmc_{config}.json" if config else "latest_pmc.json")
    for path in [first] + sorted(glob.glob(os.path.join(ROOT, "profiles", "latest_pmc_*.json"))):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        have = pmc.get("_shape")
        if shape is not None and have is not None and (set(have) != set(shape) or any(have.get(k) != v for k, v in shape.items())):
            continue
        if shape is not None and have is None and (path != first or shape != DEFAULT_SHAPES.get(config or "ivfadc")):
            continue   # (files written before the shape was recorded were taken on the default workload)
        if kernel in pmc and "fetch_kib" in pmc[kernel] and "write_kib" in pmc[kernel]:   # (both passes must have succeeded)
            return int((2 * pmc[kernel]["fetch_kib"] + pmc[kernel]["write_kib"]) * 1024)
    return None


DEFAULT_SHAPES = {"ivfadc": {"N": 3_000_000, "Q": 1024, "C": 1000, "nprobe": 10}, "pq": {"N": 1_000_000, "Q": 64},
                  "join": {"N": 1_000_000, "Q": 5000}, "exact": {"N": 3_000_000, "Q": 64}}


def roofline(kernel, avg_s, algorithmic_bytes, model, traffic, extra=None):
    ach = algorithmic_bytes / avg_s / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
         "traffic_source": ("profiles/latest_pmc*.json (rocprofv3 --pmc passes of this command on this workload shape, 2*FETCH_SIZE + WRITE_SIZE)"
                            if traffic else "no PMC pass on this workload shape"),
         "algorithmic_bytes_per_launch": int(algorithmic_bytes), "algorithmic_model": model,
         "avg_launch_us": round(avg_s * 1e6, 2)}
    if traffic:
        r["hbm_counter"] = {"achieved": round(traffic / avg_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 5),
                            "note": "counter traffic / kernel time; the counters include Infinity-Cache hits"}
    if extra:
        r.update(extra)
    return r


# ---------------------------------------------------------------------------------------------------
# What the driver reads: the LAST stdout line, kept under 1900 bytes (its tail is 2000 characters; round 4's 21 KB line was
# cut and the record lost the metric).  Everything else goes ONCE into bench_details.json beside this file (and into
# gpurun_out/ when that exists, so it comes back from the GPU box) and, as bare numbers, onto an EARLIER stdout line.
# ---------------------------------------------------------------------------------------------------
HEADLINE_MAX_BYTES = 1900
DIGEST_MAX_BYTES = 5000
_CONFIG_KEYS = ("gather_every", "gather_path", "collective", "workload", "N", "d", "C", "m", "K", "nprobe", "k", "Q", "targets", "batch", "batch_per_gpu", "parallelism", "world_size",
                "backend", "batches_in_flight", "recall_at_5", "recall_at_5_without_self", "host_abi_q1024_qps", "host_abi_q4096_qps",
                "collective_1rank_ratio")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "avg_launch_us",
              "survey_8d_step_frac", "serial_ms_per_step", "host_abi_q1024_qps")
_CPU_KEYS = ("value", "unit", "cores", "value_1_core", "kind", "sample", "parity_with_gpu_on_sample", "queries_checked")


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def headline(out):
    """The contract line: metric / value / config{workload...} / roofline / cpu_baseline and nothing that is prose."""
    if not isinstance(out, dict):
        return out
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out.get(k) for k in top}
    line["dtype"] = _short((line["dtype"] or "f32").split(" ")[0], 16)
    cfg = out.get("config") or {}
    line["config"] = {k: (_short(cfg[k], 200) if k == "workload" else cfg[k]) for k in _CONFIG_KEYS if k in cfg}
    roof = out.get("roofline")
    if roof is None:
        line["roofline"] = None
    else:
        r = {k: roof.get(k) for k in _ROOF_KEYS if k in roof}
        # where `traffic` comes from, in a word: the PMC passes are separate rocprofv3 runs of this command (committed under
        # profiles/), never counters of THIS process; null traffic = no pass on this workload shape
        r["traffic_source"] = "committed_pmc_passes:profiles/latest_pmc*.json" if roof.get("traffic") else None
        step = roof.get("step") or {}
        if "serial_ms_per_step" in step:
            r["serial_ms_per_step"] = step["serial_ms_per_step"]
        q1024 = (roof.get("host_buffer_abi") or {}).get("Q1024") or {}
        if q1024.get("queries_per_s") is not None:
            r["host_abi_q1024_qps"] = q1024["queries_per_s"]
        line["roofline"] = r
    cpu = out.get("cpu_baseline")
    if cpu is None:
        line["cpu_baseline"] = None
    else:
        c = {k: (_short(cpu[k], 120) if k == "sample" else cpu[k]) for k in _CPU_KEYS if k in cpu}
        tp = cpu.get("timed_region_parity") or {}
        if "queries_checked" in tp:
            c["queries_checked"] = tp["queries_checked"]
        line["cpu_baseline"] = c
    for k in ("dry_run", "gather_verified", "filter_bound_violations"):
        if k in out:
            line[k] = out[k]
    line["details"] = "bench_details.json"
    # never over the limit: shorten the two strings first, then drop optional keys
    for cut in (120, 60):
        if len(json.dumps(line)) <= HEADLINE_MAX_BYTES:
            break
        line["config"]["workload"] = _short(line["config"].get("workload"), cut)
        if line.get("cpu_baseline"):
            line["cpu_baseline"]["sample"] = _short(line["cpu_baseline"].get("sample"), cut)
    for k in ("details", "filter_bound_violations", "gather_verified"):
        if len(json.dumps(line)) > HEADLINE_MAX_BYTES:
            line.pop(k, None)
    if len(json.dumps(line)) > HEADLINE_MAX_BYTES:
        # last resort (never an assert: the digest and the details are already out, and a run that ends without this line is
        # recorded as no measurement at all): the contract's keys, the workload cut to 60 characters, frac and the CPU value
        roof, cpu = line.get("roofline") or {}, line.get("cpu_baseline") or {}
        line = {k: line.get(k) for k in top}
        line["config"] = {"workload": _short(cfg.get("workload"), 60)}
        line["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} if roof else None
        line["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")} if cpu else None
        if cpu:
            line["cpu_baseline"]["sample"] = _short(cpu.get("sample"), 60)
    return line


def _numbers_only(o, depth=0):
    """A digest of a measurement object: numbers, booleans and short identifiers; no prose."""
    if isinstance(o, dict):
        r = {}
        for k, v in o.items():
            if k in ("note", "loop", "sample", "algorithmic_model", "traffic_source", "survey_8d_note", "parity_scope", "measurement_order",
                     "process", "timing", "recall", "metric", "workload", "unit", "peak", "bound", "higher_is_better", "data", "dtype",
                     "vs_baseline", "scaling", "hbm_counter", "lds_gather", "per_query_model", "timed_region", "kernels", "kernels_us",
                     "kernels_overlapped", "track", "config", "flops", "index_bytes", "pin_seconds"):
                continue
            v = _numbers_only(v, depth + 1)
            if v is not None and v != {}:
                r[k] = v
        return r
    if isinstance(o, (bool, int, float)):
        return o
    if isinstance(o, str) and len(o) <= 40:
        return o
    return None


def _digest_errors(o, out, path=""):
    """error messages of failed side measurements (cut to 120 characters): a digest must say WHY a figure is missing"""
    if isinstance(o, dict):
        for k, v in o.items():
            if k == "error" and isinstance(v, str):
                out[path or "."] = _short(v, 120)
            else:
                _digest_errors(v, out, f"{path}.{k}" if path else k)
    return out


def digest(out):
    d = {"bench_digest": {k: _numbers_only(out[k]) for k in ("pipelining", "host_buffer_abi", "collective_1rank", "other_configs", "timed_region_parity")
                          if isinstance(out.get(k), dict)}}
    errs = _digest_errors({k: out.get(k) for k in ("host_buffer_abi", "other_configs", "collective_1rank", "backends")}, {})
    if errs:
        d["bench_digest"]["errors"] = errs
    for drop in ("timed_region_parity", "pipelining"):
        if len(json.dumps(d)) > DIGEST_MAX_BYTES:
            d["bench_digest"].pop(drop, None)
    if len(json.dumps(d)) > DIGEST_MAX_BYTES:   # still too long: keep value / ms_per_step / frac per config
        oc = d["bench_digest"].get("other_configs") or {}
        d["bench_digest"]["other_configs"] = {c: {k: v for k, v in o.items() if k in ("value", "ms_per_step", "error", "Q1", "Q64")}
                                              for c, o in oc.items()}
    return d if len(json.dumps(d)) <= DIGEST_MAX_BYTES else None


def emit(out, details_name="bench_details.json"):
    """Details into bench_details.json (once), a digest line, then the headline as the LAST stdout line."""
    if not isinstance(out, dict):
        print(json.dumps(out), flush=True)
        return
    out.pop("_exact", None)
    blob = json.dumps(out, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, details_name), "w") as f:
                    f.write(blob + "\n")
            except OSError as e:
                print(f"[bench] could not write {d}/bench_details.json: {e}", file=sys.stderr)
    dg = digest(out)
    if dg and dg["bench_digest"]:
        print(json.dumps(dg), flush=True)
    try:   # what C libraries still hold in stdio's buffer (RCCL prints a version banner to stdout) leaves BEFORE the contract line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(headline(out)), flush=True)


# ---------------------------------------------------------------------------------------------------
# exact brute-force kNN (k_nearest_neighbour / knn_in_exact, core_functions.c:67-81): the bench's recall
# ground truth AND a measured path of its own (other_configs.exact)
# ---------------------------------------------------------------------------------------------------
def exact_truth(x, d_qs, k, dev_index, cpu_queries=4, steps=5):
    """Pins the raw vectors (freddy_gpu_pin_vectors) and returns ([ids [Q][k] per query set], measurement dict)."""
    from freddy_amd import gpu
    N, d = x.shape
    hx = x.cpu().numpy()
    ids = np.arange(1, N + 1, dtype=np.int32)
    t0 = time.time()
    vi = gpu.VectorIndex(ids, hx, device=dev_index)
    pin_s = time.time() - t0
    h_qs = [q.cpu().numpy() for q in d_qs]
    truth = [vi.search(q, k)[0] for q in h_qs]
    info = {"metric": "exact kNN queries/sec (k_nearest_neighbour / knn_in_exact, cosine_similarity_bytea chain), 3Mx300d", "unit": "queries/s",
            "config": {"workload": f"exact brute-force kNN over {N}x{d} raw vectors, k={k}, host-buffer ABI (freddy_gpu_exact_search)",
                       "N": N, "d": d, "k": k}, "pin_seconds": round(pin_s, 2), "index_bytes": int(vi.nbytes)}
    table_bytes = N * d * 4
    for Q in (1, 64):
        hq = np.ascontiguousarray(h_qs[0][:Q])
        vi.search(hq, k)
        reps = max(3, steps)
        t0 = time.perf_counter()
        for _ in range(reps):
            got_i, got_s = vi.search(hq, k)
        dt = (time.perf_counter() - t0) / reps
        vi.profile_enable(True)
        for _ in range(3):
            vi.search(hq, k)
        prof = vi.profile_read()
        vi.profile_enable(False)
        kern = {n: round(1e3 * ms / max(l, 1), 2) for n, (l, ms) in prof.items()}
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0]
        avg_s = prof[dom][1] / max(prof[dom][0], 1) / 1e3
        flops = 2.0 * N * d * Q
        passes = (Q + 15) // 16 if Q > 8 else 1      # (a tile of queries streams the table once)
        info[f"Q{Q}"] = {
            "value": round(Q / dt, 2), "ms_per_call": round(1e3 * dt, 4), "kernels_us": kern,
            "roofline": roofline({"exact_scan": "exact_scan_kernel", "exact_filter": "exf_filter_kernel"}.get(dom, dom), avg_s, table_bytes + Q * (d * 4 + k * 8),
                                 "the raw vectors once per call (N*d*4 B = 3.6 GB: larger than the 256 MiB Infinity Cache, so this IS HBM traffic) + queries + results",
                                 pmc_traffic({"exact_scan": "exact_scan_kernel", "exact_filter": "exf_filter_kernel"}.get(dom, dom), "exact", {"N": N, "Q": Q}),
                                 {"table_passes_of_this_launch": passes,
                                        "flops": {"per_call": flops, "achieved_tflops": round(flops / avg_s / 1e12, 3),
                                                  "note": "2*N*d*Q multiply-adds of the similarity chain (separately rounded mul + add on the VALU for "
                                                          "the exact chain; the matrix cores when the filter + refine path is taken)"}})}
    # CPU oracle (port of cosine_similarity_bytea + ORDER BY ... FETCH FIRST k) on a bounded sample, and parity on it
    from oracle.oracle import Oracle
    o = Oracle()
    ns = min(cpu_queries, h_qs[0].shape[0])
    t0 = time.perf_counter()
    exp = [o.exact_knn(hx, ids, h_qs[0][j], k) for j in range(ns)]
    cdt = time.perf_counter() - t0
    gi, gs = vi.search(np.ascontiguousarray(h_qs[0][:ns]), k)
    parity = all(np.array_equal(exp[j]["id"], gi[j]) and np.array_equal(exp[j]["dist"].view(np.uint32), gs[j].view(np.uint32)) for j in range(ns))
    info["cpu_baseline"] = {"value": round(ns / cdt, 3), "unit": "queries/s", "cores": 1, "kind": "port",
                            "sample": f"the first {ns} queries of one bench batch over all {N} rows",
                            "loop": "oracle/fo_exact_knn (core_functions.c:67-81 similarity chain + ORDER BY similarity DESC FETCH FIRST k, "
                                    "freddy--0.0.1.sql:426-454), one thread; README: 8.79 s per query end to end in PostgreSQL",
                            "parity_with_gpu_on_sample": bool(parity)}
    vi.close()
    return truth, info

# ---------------------------------------------------------------------------------------------------
# config ivfadc (BASELINE configs[2]; with --gpus N: configs[4])
# ---------------------------------------------------------------------------------------------------
_BENCH_STREAMS = {}


def bench_streams(a, dev):
    """The searching streams of this process, created ONCE and before anything else creates streams (main() calls this before the
    process group exists: RCCL's own streams would otherwise take hardware queues first, and the four searching streams ended up
    sharing -- 5.9 instead of 9.4 M queries/s for the very same steps in a process that merely HAD a communicator)."""
    n_fl = max(1, min(a.in_flight, 8))
    key = (str(dev), n_fl, a.stream_skip)
    if key not in _BENCH_STREAMS:
        _skipped = [torch.cuda.Stream(dev) for _ in range(a.stream_skip)]   # (tools/sweep_queues.sh: shifts the streams' hardware queues)
        _BENCH_STREAMS[key] = (_skipped, [torch.cuda.Stream(dev) for _ in range(n_fl)])
        for st in _BENCH_STREAMS[key][1]:   # (a hardware queue is bound when a stream first runs something)
            with torch.cuda.stream(st):
                torch.zeros(1, device=dev)
        torch.cuda.synchronize(dev)
    return _BENCH_STREAMS[key][1]


def run_ivfadc(a, rank, world, dev, dev_index):
    import torch.distributed as dist
    from freddy_amd import gpu, shard, index_build as ib
    N = a.N or 3_000_000
    Q = a.Q or 1024
    if a.scaling == "strong":
        if Q % world:
            raise SystemExit("--scaling strong needs --Q divisible by the number of ranks")
        q_local = Q // world
    else:
        q_local = Q
    t0 = time.time()
    x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
    tab = ib.build_ivf_index(x, C=a.C, m=a.m, K=a.K, train_size=100000, iters=10, seed=2)
    log(rank, f"corpus+index built in {time.time() - t0:.1f}s")
    t0 = time.time()
    index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=dev_index)
    log(rank, f"pinned {index.nbytes / 1e6:.1f} MB in {time.time() - t0:.1f}s")

    # queries = indexed vectors themselves (ivfadc_batch_search takes ids), distinct per rank AND per stream: the
    # batches in flight are different batches
    n_fl = max(1, min(a.in_flight, 8))
    rng = np.random.default_rng(7 + rank)
    qids = [np.sort(rng.choice(np.arange(1, N + 1), size=q_local, replace=False)).astype(np.int64) for _ in range(n_fl)]
    if a.same_queries:
        qids = [qids[0]] * n_fl
    d_qs = [x[torch.from_numpy(q - 1).to(dev)].contiguous() for q in qids]
    d_status = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    # Everything of a step -- the search kernels (through the C ABI, on this stream's handle) and the RCCL
    # gather -- is ordered on ONE explicit non-default stream, so the collective reads a shard's results
    # only after the search wrote them and the buffer is rewritten only after the collective read it.
    # (GPU_MAX_HW_QUEUES = 8 above: a hardware queue per stream plus spares for the default stream and RCCL.)
    # Consecutive steps alternate between `in_flight` streams (each with its own query set, its own result buffer and,
    # inside the library, its own workspace): a batch is a chain of dependent kernels, and the latency-bound ends of
    # the batches overlap with the scans of the others.  --in-flight 1 is the strict sequence.
    a.warmup = max(a.warmup, n_fl)   # (every stream's workspace is allocated by its first search: never inside the timed region)
    if a.steps < n_fl:
        raise SystemExit(f"--steps must be at least --in-flight ({n_fl}): every stream's buffer is verified after the timed region")
    # (the streams are created ONCE per process: the K = 256 side configuration runs this function a second time, and four MORE
    #  streams landed on hardware queues the first four already used -- 6.1 instead of 10.4 M queries/s for that side figure)
    streams = bench_streams(a, dev)
    # the N > 1 path: every rank's lists are gathered (RCCL all_gather) -- one collective per GROUP of `gather_every` steps (default:
    # the batches in flight), a ring of two groups; --force-collective takes this path with a single rank
    collective = world > 1 or bool(a.force_collective)
    G = (a.gather_every or 1) if collective else 1   # (default: a gather per step, in the step's stream)
    depth = 2 * G if (collective and G > 1) else max(2, n_fl)
    comm, gather_path = None, ("none" if not collective else a.gather_path)
    if collective and a.gather_path == "rccl" and a.backend == "nccl":
        try:
            from freddy_amd import rccl
            # one communicator per searching stream (shard.PipelinedGather: RCCL orders the operations of ONE communicator among
            # themselves across streams); --one-comm: a single communicator for all four
            comm = rccl.Communicator(dev_index) if a.one_comm else {st.cuda_stream: rccl.Communicator(dev_index) for st in streams}
        except Exception as e:   # (the run goes on through c10d: correct, slower -- and says so in config.gather_path)
            log(rank, f"direct RCCL communicator unavailable ({type(e).__name__}: {e}); the gather goes through c10d")
            gather_path = "c10d (fallback)"
    elif collective and a.gather_path == "rccl":
        gather_path = "c10d"   # (gloo etc.)
    with torch.cuda.stream(streams[0]):
        pg = shard.PipelinedGather(q_local, a.k, dev, depth=depth, force=collective, gather_every=G, in_stream=(a.gather_path != "c10d-async"), comm=comm)
    torch.cuda.synchronize(dev)
    counter = [0]
    # world == 1: there is no collective to order, so a step is nothing but the C call -- one pre-bound ctypes call per
    # (stream, result buffer) pair (tools/burst_trace.py: torch's stream context + the gather bookkeeping cost 4 us of host
    # time per step, and the chains of a 20-step burst start that much later one after the other).  Steps and buffers both
    # advance round-robin: step c runs on stream c % n_fl and writes buffer c % depth (== the stream's own buffer: depth = n_fl).
    def step_on(n_streams, pg=pg, collective=collective):
        bound = {}
        depth = pg.depth

        def step():
            c = counter[0]
            i = c % n_streams
            st = streams[i]
            counter[0] = c + 1
            if not collective:
                b = c % depth
                pg.steps, pg.cur = c + 1, b
            else:
                # the gather's bookkeeping on THIS step's stream: wait for the gather that last read the buffer (two groups
                # back), and -- after the search -- an event, or for the group's last step the all_gather itself
                pg.next_buffer(st)
                b = pg.cur
            fn = bound.get((i, b))
            if fn is None:
                res = pg.res[b]
                fn = bound[(i, b)] = index.bind_search_dev(d_qs[i].data_ptr(), q_local, a.k, a.nprobe, 1000.0, gpu.FOUND_ROWS,
                                                           res[0].data_ptr(), res[1].data_ptr(), d_status.data_ptr(), st.cuda_stream)
            fn()
            if collective:
                pg.submit(st)
        return step

    if True:
        sync = lambda: torch.cuda.synchronize(dev)
        if collective:
            # an RCCL kernel must never queue behind 4 x 64 persistent scan workgroups that hold every CU: each scan leaves
            # CUs free (measured with a one-rank group on one GPU: other_configs.collective_1rank, profiles/r06_collective_1rank.txt)
            index.set_option("reserve_cus", a.reserve_cus)
        # ---- the side measurements FIRST: the same steps strictly one after the other, and the per-kernel durations (HIP
        # events on the launch stream; instrumented re-runs).  They are this very workload, so the timed region below starts
        # on a chip in its steady state: after an idle period (pinning the index is host work) the first ~10 ms of bursts
        # run up to 12 % slower (clocks / power state: tools/burst_trace.py RAMP=1, profiles/r04_burst_ramp.txt), which is
        # longer than the driver's whole 20-step region.
        step = step_on(n_fl)
        index.set_option("scan_share", 1)
        step1 = step_on(1)
        dt1, barrier = sharded_steps(step1, pg, a.steps, max(2, a.warmup), sync, world)
        # (at least 240 launches per kernel: averages that do not depend on --steps, and ~60 ms of this very workload before the
        # timed region -- the driver's 20-step command used to time the chip's ramp: 0.105 ms per step against 0.099 over 300 steps)
        n_prof = max(a.steps, 240)
        index.profile_enable(True)
        for _ in range(n_prof):
            step1()
        barrier()
        prof = index.profile_read()
        index.profile_enable(False)
        prof_ov = {}
        index.set_option("scan_share", a.scan_share or n_fl)   # the *_dev contract: the caller states how many batches it keeps in flight
        if n_fl > 1:   # ... and with the batches in flight as in the timed region: durations under overlap
            index.profile_enable(True)
            for _ in range(max(n_prof, 4 * n_fl)):
                step()
            barrier()
            prof_ov = index.profile_read()
            index.profile_enable(False)
        # ---- the timed region: W warmup steps, barrier, exactly K steps, barrier ----
        counter[0] = 0
        pg.steps = 0   # (drained by the barrier above: step c again writes buffer c % depth)
        dt, barrier = sharded_steps(step, pg, a.steps, a.warmup, sync, world)
        qps = world * q_local * a.steps / dt
        gather_ok = verify_gather(pg, rank, world)
        # What the TIMED region left in every stream's result buffer (steps and buffers both advance round-robin, so
        # buffer b was last written by the last timed step of stream b' = the step index modulo n_fl): kept here,
        # compared bit for bit with the oracle below -- verify what is timed.
        total_steps = counter[0]
        timed_results = []
        for back in range(min(n_fl, a.steps)):
            sidx = total_steps - 1 - back
            timed_results.append((sidx % n_fl, pg.res[sidx % depth].clone()))   # (depth is a multiple of n_fl or equal to it)
        straggler = int(d_status[0].item())
        # ---- --force-collective on one rank: the SAME process, streams and steps without the collective, alternating with the
        # collective path (what the gather, its stream and the reserved CUs cost rank 0 of an N-GPU run)
        coll_ab = None
        if a.force_collective and world == 1:
            with torch.cuda.stream(streams[0]):
                pg0 = shard.PipelinedGather(q_local, a.k, dev, depth=max(2, n_fl))
            plain = step_on(n_fl, pg0, False)
            rounds = []
            for _ in range(max(1, a.ab_rounds)):
                row = {}
                for name, fn, pgx, rc in (("without", plain, pg0, 0), ("with", step, pg, a.reserve_cus)):
                    index.set_option("reserve_cus", rc)
                    counter[0] = 0
                    pgx.steps = 0
                    for _ in range(a.warmup):
                        fn()
                    barrier_x = (lambda p=pgx: (p.drain(), sync()))
                    barrier_x()
                    t0 = time.perf_counter()
                    for _ in range(a.steps):
                        fn()
                    t_host = time.perf_counter() - t0     # the host's share: the steps are enqueued, nothing is waited for
                    barrier_x()
                    dtx = time.perf_counter() - t0
                    row[name] = round(q_local * a.steps / dtx, 1)
                    row[name + "_host_us_per_step"] = round(1e6 * t_host / a.steps, 1)
                rounds.append(row)
            index.set_option("reserve_cus", a.reserve_cus)
            w = float(np.mean([r["with"] for r in rounds])); wo = float(np.mean([r["without"] for r in rounds]))
            coll_ab = {"with_collective_qps": round(w, 1), "without_qps": round(wo, 1), "ratio": round(w / wo, 4), "rounds": rounds,
                       "gather_every": G, "gather_path": gather_path, "reserve_cus": a.reserve_cus, "backend": a.backend, "steps": a.steps,
                       "gather_verified": gather_ok,
                       "note": "one process, the same four streams: steps through the world > 1 branch (1-rank process group, the gather of "
                               "--gather-path per group of gather_every steps, option reserve_cus) against the plain --gpus 1 steps"}
        index.set_option("scan_share", 1)
        if comm is not None:
            torch.cuda.synchronize(dev)
            for cm in (comm.values() if isinstance(comm, dict) else [comm]):
                cm.close()
    scanned_rows = index.last_scanned_rows()
    n_cells, cell_rows = index.last_probed_cells()
    bound_violations = index.bound_violations()   # self-check of the filter + refine scan (must be 0)

    out = None
    if rank == 0:
        # ---- the host-buffer ABI (what pg/freddy_srf.c calls: queries in host memory, lists into host memory, one
        # synchronous call per batch; inside: pinned staging, sub-batches of up to 2048 on up to four lanes, transfers by copy
        # kernels overlapped with the neighbours' kernels): first-class numbers at three batch sizes, never `value`.
        # Measured in a CHILD process that owns nothing but the library's streams -- what a PostgreSQL backend is.  In THIS
        # process the four torch streams of the timed region were created first, and the runtime then puts the library's
        # lanes on hardware queues they share (tools/pipe_queues.py: 5.8 -> 3.6 M queries/s at 4096 queries per call).
        # The child pins the SAME tables (handed over in a file: index training uses GPU reductions whose bits can differ
        # from run to run) and compares its lists bit for bit with the lists the timed region of this process left behind.
        import tempfile
        h_sets = [q.cpu().numpy() for q in d_qs]
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            path = os.path.join(td, "sets.npz")
            np.savez(path, **{f"tab_{n}": np.asarray(tab[n]) for n in ("coarse", "codebook", "list_off", "ids", "codes")},
                     **{f"q{i}": h_sets[i] for i in range(n_fl)},
                     **{f"ids{i}": r[0].cpu().numpy() for i, r in timed_results},
                     **{f"dist{i}": r[1].cpu().numpy() for i, r in timed_results})
            cmd = [sys.executable, os.path.abspath(__file__), "--host-abi-child", path, "--N", str(N), "--C", str(a.C), "--m", str(a.m),
                   "--K", str(a.K), "--nprobe", str(a.nprobe), "--k", str(a.k)]
            try:
                if a.no_host_abi:
                    raise RuntimeError("skipped (--no-host-abi)")
                if world > 1:   # (a side measurement of ONE backend's call: reported at N = 1, like the CPU baseline; the other ranks would wait for it)
                    raise RuntimeError("skipped (reported at --gpus 1 only)")
                # (ONE backend that has the GPU to itself: six hardware queues, what the library picks for a process that finds no
                # other live backend -- this parent, with its pinned index, would count as one)
                cp = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, GPU_MAX_HW_QUEUES="6"))
                host_abi = json.loads(cp.stdout.strip().splitlines()[-1])
            except Exception as e:   # the headline line must not be lost to a side measurement
                host_abi = {"error": f"{type(e).__name__}: {e}"}
        host_qps = (host_abi.get(f"Q{q_local}") or {}).get("queries_per_s")
        host_same = all(v.get("same_results_as_device_path", True) for v in host_abi.values() if isinstance(v, dict)) and "error" not in host_abi

        row_bytes = a.m * (1 if a.K <= 256 else 2) + 4   # (K <= 256: one byte per code, packed8 -- SURVEY 8d's cb = 1)
        per_query_bytes = scanned_rows * row_bytes + q_local * (300 * 4 + a.k * 8)          # SURVEY 8d, per query
        shared = a.C * 300 * 4 + a.m * a.K * (300 // a.m) * 4                                # coarse + codebook, once
        cell_bytes = cell_rows * row_bytes + q_local * (300 * 4 + a.k * 8) + shared           # every probed list ONCE
        kern = {n: {"launches": l, "avg_us": round(1e3 * ms / max(l, 1), 2)} for n, (l, ms) in prof.items()}
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0] if prof else None
        roof = None
        if dom:
            avg_s = prof[dom][1] / max(prof[dom][0], 1) / 1e3
            scan_variant = os.environ.get("FREDDY_GPU_FUSED_KERNEL", "5")
            # K <= 256 with one byte per code (option codes_u8 = 1, the default) takes the whole-entry-slab kernel (fused8.h)
            filt = "ivf_filter8_kernel" if (a.K <= 256 and os.environ.get("FREDDY_GPU_CODES_U8", "1") == "1") else "ivf_filter5_kernel"
            kname = {"ivf_filter": filt if scan_variant == "5" else "ivf_filter_kernel", "ivf_exact_scan": "ivf_spec2_kernel", "adc_scan": "adc_scan_kernel",
                     "lut_build": "lut_build_kernel", "coarse_dist": "coarse_tile_kernel",
                     "probe_plan": "probe_plan_kernel"}.get(dom, dom)
            shape_now = {"N": N, "Q": q_local, "C": a.C, "nprobe": a.nprobe}
            if a.K != 1024:
                shape_now["K"] = a.K   # (a PMC record of the K = 1024 workload does not describe this one)
            traffic = pmc_traffic(kname, None, shape_now)
            ov_dom = prof_ov.get(dom)
            if "sparse_items" in prof and "ivf_filter" in prof:
                # the scan is two launches here (thin cells item by item, the others cell-grouped): the algorithmic bytes are
                # those of BOTH, so both durations and both kernels' counters are priced together
                dom = "ivf_filter+sparse_items"
                # (thin cells: units of two queries by default -- option sparse_items >= 2, sparse5.h)
                sparse = "sparse_item5_kernel" if os.environ.get("FREDDY_GPU_SPARSE_ITEMS", "2") in ("0", "1") else "sparse_pair5_kernel"
                kname = f"{filt} + {sparse}"
                avg_s = sum(prof[n][1] / max(prof[n][0], 1) for n in ("ivf_filter", "sparse_items")) / 1e3
                tr = [pmc_traffic(n, None, shape_now) for n in (filt, sparse)]
                traffic = sum(tr) if all(t is not None for t in tr) else None
                if all(n in prof_ov for n in ("ivf_filter", "sparse_items")):
                    ov_dom = (1, sum(prof_ov[n][1] / max(prof_ov[n][0], 1) for n in ("ivf_filter", "sparse_items")))
                else:
                    ov_dom = None
            slab_b = 2 if scan_variant == "5" else 4
            lds_bytes = scanned_rows * a.m * slab_b   # one table value per (query, probed row, position): int16 (fused5.h) / fp32
            lds = lds_bytes / avg_s / 1e9
            pq_ach = per_query_bytes / avg_s / 1e9
            ceiling_qps = HBM_PEAK_GBS * 1e9 / (per_query_bytes / q_local)
            roof = roofline(
                kname, avg_s, cell_bytes,
                "cell-grouped, as the reference's own loop (freddy.c:939-974 reads a probed cell's rows once per round and "
                f"offers each to every query of the cell): {row_bytes} B per row of every DISTINCT probed list + queries + results + "
                "coarse and codebook tables once",
                traffic,
                {"lds_gather": {"achieved": round(lds, 1), "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": round(lds / LDS_PEAK_GBS, 5),
                                "bytes_per_launch": int(lds_bytes),
                                "note": f"the resource that binds this kernel's main loop: {slab_b} B of slab per (query, probed row, "
                                        "position) gathered from LDS with one 16-byte read per 8 items at random rows (~1.9 "
                                        "conflict passes per read after the pin-time row arrangement; the cost is per access, "
                                        "not per byte: DESIGN.md 5.3c)"},
                 "per_query_model": {"bytes_per_launch": int(per_query_bytes), "achieved": round(pq_ach, 1), "unit": "GB/s",
                                     "note": "SURVEY 8d's per-QUERY bytes (every probed row once per query) / kernel time: an "
                                             "equivalent rate, not traffic -- the kernel reads a list once per work entry",
                                     "step_qps_ceiling_at_8TBs": round(ceiling_qps, 1),
                                     "step_frac_of_ceiling": round(qps / world / ceiling_qps, 5)},
                 "timed_region": (lambda ov: None if not ov else {
                     "avg_launch_us": round(1e3 * ov[1] / max(ov[0], 1), 2), "batches_in_flight": n_fl,
                     "share_of_chip": round(1.0 / n_fl, 4),
                     "achieved": round(cell_bytes / (ov[1] / max(ov[0], 1) / 1e3) / 1e9, 2), "unit": "GB/s",
                     "frac_of_share": round(cell_bytes / (ov[1] / max(ov[0], 1) / 1e3) / 1e9 / (HBM_PEAK_GBS / n_fl), 5),
                     "note": "the same kernel inside the timed region: with n batches in flight a scan runs on n_cus / n "
                             "workgroups beside the scans of the other batches (DESIGN.md 5.2c); frac_of_share prices it against "
                             "that share of the HBM peak, the headline frac above prices the kernel alone on every CU"})(ov_dom),
                 "distinct_probed_cells": int(n_cells), "index_bytes": int(index.nbytes),
                 "note": (f"the {index.nbytes / 1e6:.0f} MB index is Infinity-Cache (256 MiB) resident after first touch; a "
                          "non-resident corpus (N = 40 M) is measured in profiles/ (DESIGN.md 8)") if index.nbytes < (256 << 20)
                         else f"the {index.nbytes / 1e6:.0f} MB index does not fit the 256 MiB Infinity Cache: the lists come from HBM"})

        # ---- recall@5 vs exact search (over every stream's batch of the TIMED region) -------------------------
        recall = None
        recall_info = None
        other_exact = None
        if not a.no_recall:
            # ground truth = the product's own exact brute-force kNN (freddy_gpu_exact_search) over the raw
            # vectors: rows are L2-normalised, so ORDER BY cosine similarity DESC is ORDER BY L2 distance ASC; cross-checked
            # on one batch against the torch matmul ground truth this file used until round 3
            truth, exact_info = exact_truth(x, d_qs, a.k, dev_index)
            rec = [ib.recall_at_k(r[0].cpu().numpy(), truth[i]) for i, r in timed_results]
            recall = float(np.mean(rec))
            # ... and over the OTHER k - 1 neighbours: a query is an indexed row, so its own id leads the exact list
            def _without_self(res_ids, exact_ids, self_ids):
                hits = n = 0
                for a, e, me in zip(res_ids, exact_ids, self_ids):
                    es = set(int(v) for v in e) - {int(me)}
                    hits += len((set(int(v) for v in a if v >= 0) - {int(me)}) & es)
                    n += len(es)
                return hits / float(max(n, 1))
            recall_wo = float(np.mean([_without_self(r[0].cpu().numpy(), truth[i], qids[i]) for i, r in timed_results]))
            i0, r0 = timed_results[0]
            rec_torch = ib.recall_at_k(r0[0].cpu().numpy(), ib.exact_topk(x, d_qs[i0], a.k))
            recall_info = {"ground_truth": "freddy_gpu_exact_search (exact.h) over the 3 M raw vectors, k = 5, every query of the timed region's batches",
                           "cross_check_torch_matmul_one_batch": {"recall_exact_kernel": round(ib.recall_at_k(r0[0].cpu().numpy(), truth[i0]), 5),
                                                                  "recall_torch": round(rec_torch, 5)},
                           "includes_self_match": True, "recall_at_5_without_self": round(recall_wo, 4),
                           "note": "queries are indexed rows (ivfadc_batch_search takes ids, freddy.c:679-999): rank 1 of the exact list is "
                                   "the query itself, so 0.2 of the value is the self-match (found whenever the query's own cell is probed)"}
            other_exact = exact_info

        # ---- CPU oracle: timed on a bounded sample; EVERY list the timed region left behind compared bit for bit ----
        cpu = None
        timed_parity = None
        if a.cpu_sample > 0 and world == 1:   # reported at N=1 only
            from oracle.oracle import Oracle
            o = Oracle()
            ot = o.ivf_table(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"])
            ns = min(a.cpu_sample, q_local)
            cores = os.cpu_count() or 1
            timed_parity = {"buffers_checked": 0, "queries_checked": 0, "all_equal": True}
            cdt = None
            for i, r in sorted(timed_results):
                nq = q_local if a.cpu_sample >= q_local else ns
                qs = d_qs[i][:nq].cpu().numpy()
                t0 = time.perf_counter()
                exp = o.ivfadc_search_many(ot, qs, a.k, a.nprobe, sentinel=1000.0, found_rule=0, n_threads=cores)
                if cdt is None:
                    cdt, ns = time.perf_counter() - t0, nq
                ok = bool(np.array_equal(exp["id"], r[0][:nq].cpu().numpy()) and
                          np.array_equal(exp["dist"].view(np.uint32), r[1][:nq].cpu().numpy().view(np.uint32)))
                timed_parity["buffers_checked"] += 1
                timed_parity["queries_checked"] += nq
                timed_parity["all_equal"] = timed_parity["all_equal"] and ok
            parity = timed_parity["all_equal"]
            n1 = min(64, ns)   # one thread = one PostgreSQL backend
            t0 = time.perf_counter()
            o.ivfadc_search_many(ot, d_qs[0][:n1].cpu().numpy(), a.k, a.nprobe, sentinel=1000.0, found_rule=0, n_threads=1)
            one_core = n1 / (time.perf_counter() - t0)
            cpu = {"value": round(ns / cdt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                   "value_1_core": round(one_core, 2),
                   "sample": f"the first {ns} queries of one of the {n_fl} bench batches ({q_local} queries each), same index, nprobe={a.nprobe}, k={a.k}",
                   "loop": "oracle/fo_ivfadc_search_many: per query the W-probe loop of ivfadc_search (freddy.c:174-393: W best "
                           "cells, W LUTs, rows of the W lists merged by id, updateTopK), OpenMP over queries = one backend per "
                           "core; gcc -O2 without -march=native (PGXS defaults).  SPI / tuple / per-call table reload cost of the "
                           "real UDF is NOT included (README: ~0.01 s per query end to end)",
                   "parity_with_gpu_on_sample": parity,
                   "parity_scope": f"the result buffer every one of the {n_fl} in-flight streams held when the TIMED region ended "
                                   f"({timed_parity['queries_checked']} queries, {n_fl} different batches), ids and distance bits"}

        serial_ms = round(1e3 * dt1 / a.steps, 4)
        abi_brief = {n: {"queries_per_s": v.get("queries_per_s"), "ms_per_call": v.get("ms_per_call"),
                         "same_results_as_device_path": v.get("same_results_as_device_path")}
                     for n, v in host_abi.items() if isinstance(v, dict) and n.startswith("Q")}
        # The objects the driver keeps whole (config / roofline / cpu_baseline) carry every number the README quotes:
        # recall, parity of the timed region, the host-buffer ABI, the serial step, SURVEY 8d's per-query fraction.
        if roof is not None:
            roof["survey_8d_step_frac"] = roof["per_query_model"]["step_frac_of_ceiling"]
            roof["survey_8d_note"] = ("whole-step rate in SURVEY 8d's per-QUERY bytes (every probed row once per query) / 8 TB/s; can "
                                      "approach or exceed 1 because a list is read once for up to 16 queries of its cell, as freddy.c:939-974 does")
            roof["step"] = {"ms_per_step": round(1e3 * dt / a.steps, 4), "serial_ms_per_step": serial_ms, "batches_in_flight": n_fl}
            roof["host_abi_q1024_qps"] = (abi_brief.get("Q1024") or {}).get("queries_per_s")   # (PCIe-inclusive; never `value`; all sizes: host_buffer_abi)
        if cpu is not None:
            cpu["timed_region_parity"] = timed_parity
            cpu["filter_bound_violations"] = bound_violations
        out = {
            "metric": METRIC,
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32 (results: the reference's binary32 sub/mul/add chains; filter stage: int16 fixed-point table x f32 scale, f32 sums)",
            "data": "synthetic",
            "config": {"workload": "IVFADC batch (ivfadc_batch_search generalised to nprobe), "
                                   f"{N}x300d, C={a.C}, m={a.m}, K={a.K}, nprobe={a.nprobe}, k={a.k}, "
                                   f"batch={q_local} queries per GPU, replicated index, queries sharded by rank",
                       "N": N, "d": 300, "C": a.C, "m": a.m, "K": a.K, "nprobe": a.nprobe, "k": a.k,
                       "batch_per_gpu": q_local, "parallelism": f"dp{world}",
                       "batches_in_flight": n_fl, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "world_size": (dist.get_world_size() if dist.is_initialized() else 1),
                       "backend": (("rccl (torch.distributed nccl)" if a.backend == "nccl" else a.backend) if collective else "none (single GPU)"),
                       "gather_every": (G if collective else None), "gather_path": gather_path,
                       "communicators": (len(comm) if isinstance(comm, dict) else (1 if comm is not None else None)),
                       "recall_at_5": None if recall is None else round(recall, 4),
                       "recall_at_5_without_self": None if recall_info is None else recall_info["recall_at_5_without_self"],
                       "host_abi_q1024_qps": (abi_brief.get("Q1024") or {}).get("queries_per_s"),
                       "host_abi_q4096_qps": (abi_brief.get("Q4096") or {}).get("queries_per_s"),
                       "recall": recall_info,
                       "queries_needing_extra_round": straggler,
                       "measurement_order": "serial and instrumented passes of the same steps first, then W warmup steps, barrier, the K timed steps, "
                                            "barrier (a chip that idled while the index was pinned runs its first ~10 ms of bursts up to 12 % "
                                            "slower: profiles/r04_burst_ramp.txt)"},
            "recall_at_5": None if recall is None else round(recall, 4),
            "queries_needing_extra_round": straggler,
            "filter_bound_violations": bound_violations,
            "gather_verified": gather_ok,
            "pipelining": {"batches_in_flight": n_fl, "serial_ms_per_step": serial_ms,
                           "serial_queries_per_s": round(world * q_local * a.steps / dt1, 1),
                           "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                           "note": "value = throughput with consecutive batches (different query sets) on alternating HIP "
                                   "streams, option scan_share = batches in flight (each batch's persistent scan takes n_cus / "
                                   "scan_share workgroups so the scans run side by side); serial_* = the same steps strictly one "
                                   "after the other (scan_share = 1: the scan takes every CU)"},
            "timed_region_parity": timed_parity,
            "collective_1rank": coll_ab,
            "host_buffer_abi": dict(host_abi, queries_per_s=host_qps, same_results_as_device_path=host_same,
                                    note="freddy_gpu_ivfadc_search, the call the PostgreSQL hosts make (pageable host buffers in and "
                                         "out, synchronous): sub-batches of up to 2048 queries on up to four library-owned lanes with pinned "
                                         "staging, H2D / D2H overlapped with the neighbours' kernels, extra probing rounds where the "
                                         "host waits for a lane; queries_per_s = the 1024-query call"),
            "roofline": roof, "kernels": kern,
            "kernels_overlapped": {n: {"launches": l, "avg_us": round(1e3 * ms / max(l, 1), 2)} for n, (l, ms) in prof_ov.items()},
            "cpu_baseline": cpu,
        }
        if other_exact is not None:
            out["_exact"] = other_exact
    return out


# ---------------------------------------------------------------------------------------------------
# config pq (BASELINE configs[1]): pq_search, 1M x 300d, m=12, K=1024, k=5
# ---------------------------------------------------------------------------------------------------
def run_pq(a, rank, world, dev, dev_index):
    from freddy_amd import gpu, index_build as ib
    N = a.N or 1_000_000
    Q = a.Q or 64
    x = ib.make_corpus(N, d=300, seed=11, device=dev)
    tab = ib.build_pq_index(x, m=a.m, K=a.K, train_size=100000, iters=6, seed=1)
    index = gpu.PQIndex(tab["codebook"], tab["ids"], tab["codes"], device=dev_index)
    rng = np.random.default_rng(7 + rank)
    qids = np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False)).astype(np.int64)
    d_q = x[torch.from_numpy(qids - 1).to(dev)].contiguous()
    d_ids = torch.empty((Q, a.k), dtype=torch.int32, device=dev)
    d_dist = torch.empty((Q, a.k), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    stream = torch.cuda.Stream(dev)

    def step():
        index.search_dev(d_q.data_ptr(), Q, a.k, 100.0, d_ids.data_ptr(), d_dist.data_ptr(), stream.cuda_stream)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    index.profile_enable(True)
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(dev)
    prof = index.profile_read()
    index.profile_enable(False)
    kern = {n: {"launches": l, "avg_us": round(1e3 * ms / max(l, 1), 2)} for n, (l, ms) in prof.items()}
    # batches of >= 16 queries take the cell-grouped filter + refine scan over pseudo-lists; smaller ones adc_scan_kernel
    dom = "ivf_filter" if "ivf_filter" in prof else "adc_scan"
    dom_kernel = "ivf_filter5_kernel" if dom == "ivf_filter" else "adc_scan_kernel"
    avg_s = prof[dom][1] / max(prof[dom][0], 1) / 1e3
    row_bytes = a.m * 2 + 4
    once = N * row_bytes + Q * (a.m * a.K * 4 + 300 * 4 + a.k * 8)   # the table once + every query's LUT
    per_query = Q * (N * row_bytes + a.m * a.K * 4)
    roof = roofline(dom_kernel, avg_s, once,
                    "the code table once per batch (28 B per row) + one 48 KiB LUT, query and result per query",
                    pmc_traffic(dom_kernel, "pq", {"N": N, "Q": Q}),
                    {"per_query_model": {"bytes_per_launch": int(per_query), "achieved": round(per_query / avg_s / 1e9, 1), "unit": "GB/s",
                                         "note": "SURVEY 8d: N*(m*2+4) = 28 MB per QUERY / kernel time (an equivalent rate: the "
                                                 "cell-grouped scan reads a 4096-row pseudo-list once per 16 queries; adc_scan_kernel, "
                                                 "for batches below 16 queries, re-reads the table from the caches for every query)"},
                     "lds_gather": {"achieved": round(Q * N * a.m * (2 if dom == "ivf_filter" else 4) / avg_s / 1e9, 1), "peak": LDS_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(Q * N * a.m * (2 if dom == "ivf_filter" else 4) / avg_s / 1e9 / LDS_PEAK_GBS, 5)}})
    from oracle.oracle import Oracle
    o = Oracle()
    ot = o.pq_table(tab["codebook"], tab["ids"], tab["codes"])
    qs = d_q.cpu().numpy()
    ns = min(Q, 32)
    t0 = time.perf_counter()
    exp = np.stack([o.pq_search(ot, q, a.k) for q in qs[:ns]])
    cdt = time.perf_counter() - t0
    parity = bool(np.array_equal(exp["id"], d_ids[:ns].cpu().numpy()) and
                  np.array_equal(exp["dist"].view(np.uint32), d_dist[:ns].cpu().numpy().view(np.uint32)))
    h_q = qs[:1]
    one_call, one_i, one_d = index.bind_search(h_q, a.k, sentinel=100.0)   # (arguments converted once: a C caller's cost)
    one_call()
    t0 = time.perf_counter()
    for _ in range(50):
        one_call()
    one_ms = (time.perf_counter() - t0) / 50 * 1e3
    single_parity = bool(np.array_equal(one_i[0], exp["id"][0]) and np.array_equal(one_d[0].view(np.uint32), exp["dist"][0].view(np.uint32)))
    # the reference's own call shape -- pq_search(bytea, int), freddy.c:28-152: ONE query -- with its kernels and a roofline
    index.profile_enable(True)
    for _ in range(10):
        index.search(h_q, a.k, sentinel=100.0)
    prof1 = index.profile_read()
    index.profile_enable(False)
    kern1 = {n: round(1e3 * ms / max(l, 1), 2) for n, (l, ms) in prof1.items()}
    single = {"ms_per_call": round(one_ms, 4), "parity_with_oracle": single_parity, "kernels_us": kern1, "kernels_sum_us": round(sum(kern1.values()), 2),
              "note": "host-buffer call; one query = ONE launch (pq_one_kernel: table slices, grid barrier, scan, last-arriver merge), the host polls the kernel's completion word"}
    if "adc_scan" in prof1:
        t1 = prof1["adc_scan"][1] / max(prof1["adc_scan"][0], 1) / 1e3
        single["roofline"] = roofline("adc_scan_kernel", t1, N * row_bytes + a.m * a.K * 4, "the code table once (28 B per row) + the query's 48 KiB LUT",
                                      None, {"note": "the table is Infinity-Cache resident; the call is a chain of latency-bound launches around this kernel"})
    return {
        "metric": "PQ search queries/sec (pq_search, k=5), 1Mx300d", "value": round(Q * a.steps / dt, 1), "unit": "queries/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PQ search (pq_search / knn_in_pq), {N}x300d, m={a.m}, K={a.K}, k={a.k}, batch={Q} queries",
                   "N": N, "m": a.m, "K": a.K, "k": a.k, "batch": Q},
        "single_query_host_abi_ms": round(one_ms, 4), "single_query": single,
        "roofline": roof, "kernels": kern,
        "cpu_baseline": {"value": round(ns / cdt, 2), "unit": "queries/s", "cores": 1, "kind": "port",
                         "sample": f"first {ns} bench queries, same table",
                         "loop": "oracle/fo_pq_search (freddy.c:28-152: LUT, ADC over all rows, updateTopK), one thread; SPI excluded",
                         "parity_with_gpu_on_sample": parity}}


# ---------------------------------------------------------------------------------------------------
# config join (BASELINE configs[3]): knn_join 5,000 x 100,000, k=5, alpha=100, pvf=20, method 2
# ---------------------------------------------------------------------------------------------------
def run_join(a, rank, world, dev, dev_index):
    from freddy_amd import gpu, index_build as ib
    N = a.N or 1_000_000
    Q = a.Q or 5000
    T = 100_000
    x = ib.make_corpus(N, d=300, seed=5, device=dev)
    t = ib.build_ivpq_index(x, m=30, K=32, k_coarse=32, train_size=100000, iters=6, seed=3)
    index = gpu.IVPQIndex(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"], device=dev_index)
    rng = np.random.default_rng(4)
    qid = rng.choice(np.arange(1, N + 1), Q, replace=False)
    # THREE different target arrays, taken in turn: the library keeps the resolution of the last call's "fq.id IN (targets)"
    # (join.h: mark / offsets / place kernels + the 400 KB upload) and a call with the same array finds its buckets in place.
    # The reference resolves the array on every call (ivpq_search_in.c:357-395), so `value` is the COLD figure -- no step of the
    # timed region repeats its predecessor's array; the repeated-array figure is reported beside it (cached_targets).
    target_sets = [rng.choice(np.arange(1, N + 1), T, replace=False).astype(np.int32) for _ in range(3)]
    qs = t["vectors"][qid - 1]
    alpha, pvf, method = 100, 20, 2
    n_call = [0]

    def call(cached=False):
        tg = target_sets[0] if cached else target_sets[n_call[0] % 3]
        n_call[0] += 1
        return (tg,) + tuple(index.knn_join(qs, a.k, tg, alpha, pvf, method))

    for _ in range(max(a.warmup, 1)):
        call()
    track = None
    step_s = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t1 = time.perf_counter()
        targets, gi, gd, it = call()
        step_s.append(time.perf_counter() - t1)
        tr = index.last_track()
        track = tr if track is None else {n: track[n] + tr[n] for n in tr}
    dt = time.perf_counter() - t0
    med = float(np.median(step_s))   # (a host hiccup -- one 66 ms step among twenty of 0.9 ms was seen in the combined run -- must not set the median)
    mean = dt / a.steps
    call(cached=True)
    call(cached=True)
    cached_s = []
    for _ in range(a.steps):
        t1 = time.perf_counter()
        call(cached=True)
        cached_s.append(time.perf_counter() - t1)
    # the query buffer in pinned memory (freddy_gpu_host_alloc: what pg/freddy_gpu_glue.c's query_buffer() hands over): read by the
    # kernels where it is, no staging copy; target arrays in turn as above
    pb = gpu.PinnedBuffer(qs.shape)
    pb.array[:] = qs
    pinned_s = []
    pin_ok = True
    for i in range(a.steps + 2):
        tg = target_sets[n_call[0] % 3]
        n_call[0] += 1
        t1 = time.perf_counter()
        pi, pd, pit = index.knn_join(pb.array, a.k, tg, alpha, pvf, method)
        if i >= 2:
            pinned_s.append(time.perf_counter() - t1)
    # (the last pinned call used target_sets[(n_call - 1) % 3]: compared with a pageable call on the same array)
    ci, cd, cit = index.knn_join(qs, a.k, tg, alpha, pvf, method)
    pin_ok = bool(np.array_equal(pi, ci) and np.array_equal(pd.view(np.uint32), cd.view(np.uint32)) and pit == cit)
    pb.close()
    track = {n: (v / a.steps) for n, v in track.items()}
    kernel_s = track["join_kernel_time"]
    rows = track["candidate_rows"]
    alg = rows * (30 * 2 + 4) + Q * (a.k * pvf * 300 * 4 + 300 * 4 + a.k * 8)   # codes + ids, PV vectors, query, result
    # (the PMC record is per DISPATCH of the kernel; a call makes one dispatch per alpha round: both sides per CALL here)
    per_dispatch = pmc_traffic("join_query_kernel", "join", {"N": N, "Q": Q})
    n_disp = max(1, int(round(track["iterations"])))
    roof = roofline("join_query_kernel", kernel_s, alg,
                    "SURVEY 8d: sum over queries of the target rows in their cells x (m*2+4) B + k*pvf PV vectors (1200 B each) + query + result",
                    None if per_dispatch is None else per_dispatch * n_disp,
                    {"candidate_rows_per_call": int(rows), "iterations": track["iterations"], "dispatches_per_call": n_disp,
                     "traffic_per_dispatch": per_dispatch,
                     "note": "the call is a host loop (alpha doubling, multi-index traversal in libm on the host cores): "
                             "the kernel is " + f"{100 * kernel_s / mean:.0f} % of a call"})
    from oracle.oracle import Oracle
    o = Oracle()
    ot = o.ivpq_table(t["codebook"], t["coarse"], t["ids"], t["coarse_id"], t["codes"], t["vectors"], t["stats"])
    t0 = time.perf_counter()
    exp, eit = o.ivpq_search_in(ot, qs, a.k, targets, alpha, pvf, method)
    cdt = time.perf_counter() - t0
    parity = bool(np.array_equal(exp["id"].reshape(gi.shape), gi) and
                  np.array_equal(exp["dist"].reshape(gd.shape).view(np.uint32), gd.view(np.uint32)) and eit == it)
    return {
        "metric": "kNN-join queries/sec (ivpq_search_in, 5000 x 100000, k=5, alpha=100, pvf=20, method 2)",
        "value": round(Q / mean, 1), "unit": "queries/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * mean, 4), "median_ms_per_step": round(1e3 * med, 4), "slowest_step_ms": round(1e3 * max(step_s), 3),
        "cached_targets": {"mean_ms_per_step": round(1e3 * float(np.mean(cached_s)), 4), "median_ms_per_step": round(1e3 * float(np.median(cached_s)), 4),
                           "note": "every call repeats ONE target array: the library finds the array's buckets in place (not what the reference does per call)"},
        "pinned_queries": {"mean_ms_per_step": round(1e3 * float(np.mean(pinned_s)), 4), "median_ms_per_step": round(1e3 * float(np.median(pinned_s)), 4),
                           "queries_per_s": round(Q / float(np.mean(pinned_s)), 1), "same_results_as_pageable": pin_ok,
                           "note": "the query batch in a freddy_gpu_host_alloc buffer (as pg/freddy_gpu_glue.c passes it): no staging copy; cold target arrays"},
        "timing": "value / ms_per_step = total time of the timed region / steps (mean), one synchronous call per step, no step with the target "
                  "array of its predecessor (three arrays in turn: the resolution of fq.id IN (targets) is inside every step); median beside it",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"knn_join (ivpq_search_in): {Q} queries x {T} targets of {N} rows, k={a.k}, alpha={alpha}, "
                               f"pvf={pvf}, method=2, host-buffer ABI (one synchronous call per step)",
                   "N": N, "Q": Q, "targets": T, "m": 30, "K": 32, "coarse": "2 x 32"},
        "track": {n: (round(v, 6) if isinstance(v, float) else v) for n, v in track.items()},
        "roofline": roof,
        "cpu_baseline": {"value": round(Q / cdt, 2), "unit": "queries/s", "cores": 1, "kind": "port",
                         "sample": f"the same call ({Q} queries, {T} targets), once",
                         "loop": "oracle/fo_ivpq_search_in (ivpq_search_in.c:61-699), one thread; SPI / SQL string building excluded "
                                 "(README: 2.7-20 s for this join end to end)",
                         "parity_with_gpu_on_sample": parity}}


def run_exact(a, rank, world, dev, dev_index):
    """--config exact: brute-force kNN over the raw vectors (k_nearest_neighbour / knn_in_exact) alone -- for the rocprofv3 passes."""
    from freddy_amd import index_build as ib
    N = a.N or 3_000_000
    Q = a.Q or 64
    x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
    rng = np.random.default_rng(7)
    qids = np.sort(rng.choice(np.arange(1, N + 1), size=max(Q, 64), replace=False)).astype(np.int64)
    d_q = x[torch.from_numpy(qids - 1).to(dev)].contiguous()
    _, info = exact_truth(x, [d_q], a.k, dev_index, steps=a.steps)
    key = f"Q{Q}" if f"Q{Q}" in info else "Q64"
    out = {"metric": info["metric"], "value": info[key]["value"], "unit": "queries/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": info[key]["ms_per_call"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "config": info["config"], "roofline": info[key]["roofline"], "cpu_baseline": info["cpu_baseline"],
           "kernels": info[key]["kernels_us"], "Q1": info["Q1"], "Q64": info["Q64"]}
    return out


def run_host_abi_child(a):
    """bench.py --host-abi-child sets.npz: the host-buffer ABI in a process of its own (see run_ivfadc).  Prints one JSON object."""
    from freddy_amd import gpu, index_build as ib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    z = np.load(a.host_abi_child)
    index = gpu.IVFIndex(z["tab_coarse"], z["tab_codebook"], z["tab_list_off"], z["tab_ids"], z["tab_codes"], device=0)
    n_fl = len([n for n in z.files if n.startswith("q")])
    h_sets = [z[f"q{i}"] for i in range(n_fl)]
    q_local = h_sets[0].shape[0]
    host_abi = {}
    for mult in (1, 4, 8):
        hq = np.ascontiguousarray(np.concatenate([h_sets[j % n_fl] for j in range(mult)]))
        hi, hd = index.search(hq, a.k, a.nprobe)
        reps = max(3, 24 // mult)
        t0 = time.perf_counter()
        for _ in range(reps):
            hi, hd = index.search(hq, a.k, a.nprobe)
        hdt = (time.perf_counter() - t0) / reps
        same = True
        for j in range(mult):   # the same bits as the device-resident lists of that query set in the parent's timed region
            i = j % n_fl
            if f"ids{i}" not in z.files:
                continue
            blk = slice(j * q_local, (j + 1) * q_local)
            same = same and bool(np.array_equal(hi[blk], z[f"ids{i}"]) and np.array_equal(hd[blk].view(np.uint32), z[f"dist{i}"].view(np.uint32)))
        host_abi[f"Q{mult * q_local}"] = {"queries_per_s": round(mult * q_local / hdt, 1), "ms_per_call": round(1e3 * hdt, 4),
                                         "same_results_as_device_path": same}
    one = h_sets[0][:1]
    index.search(one, a.k, a.nprobe)
    t0 = time.perf_counter()
    for _ in range(50):
        index.search(one, a.k, a.nprobe)
    host_abi["Q1"] = {"ms_per_call": round((time.perf_counter() - t0) / 50 * 1e3, 4),
                      "note": "ONE query per call: the shape of the reference's ivfadc_search(bytea, int) SRF (freddy.c:174-393)"}
    index.profile_enable(True)
    for _ in range(10):
        index.search(one, a.k, a.nprobe)
    prof1 = index.profile_read()
    index.profile_enable(False)
    kern1 = {n: round(1e3 * ms / max(l, 1), 2) for n, (l, ms) in prof1.items()}
    host_abi["Q1"]["kernels_us"] = kern1
    host_abi["Q1"]["kernels_sum_us"] = round(sum(kern1.values()), 2)
    if "adc_scan" in prof1:
        n_rows = int(z["tab_ids"].shape[0])
        n_cells = int(z["tab_list_off"].shape[0]) - 1
        byt = a.nprobe * (n_rows / max(n_cells, 1)) * (z["tab_codes"].shape[1] * 2 + 4) + a.nprobe * z["tab_codes"].shape[1] * a.K * 4
        t1 = prof1["adc_scan"][1] / max(prof1["adc_scan"][0], 1) / 1e3
        host_abi["Q1"]["roofline"] = {"bound": "hbm", "kernel": "adc_scan_kernel", "achieved": round(byt / t1 / 1e9, 2), "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round(byt / t1 / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                                      "algorithmic_bytes_per_launch": int(byt), "avg_launch_us": round(t1 * 1e6, 2),
                                      "algorithmic_model": "nprobe lists of mean length (28 B per row) + nprobe LUTs of 48 KiB",
                                      "note": "under a megabyte per query: the call is a chain of latency-bound launches, not a bandwidth problem"}
    pb = gpu.PinnedBuffer((4 * q_local, 300))
    pb.array[:] = np.concatenate([h_sets[j % n_fl] for j in range(4)])
    index.search(pb.array, a.k, a.nprobe)
    t0 = time.perf_counter()
    for _ in range(6):
        index.search(pb.array, a.k, a.nprobe)
    host_abi[f"Q{4 * q_local}_pinned_queries"] = {"queries_per_s": round(6 * 4 * q_local / (time.perf_counter() - t0), 1),
                                                   "note": "queries written into a freddy_gpu_host_alloc buffer: no staging copy"}
    pb.close()
    host_abi["process"] = "child process owning only the library's streams (a PostgreSQL backend's situation)"
    print(json.dumps(host_abi), flush=True)
    return 0


def main():
    a = parse()
    if a.host_abi_child:
        sys.exit(run_host_abi_child(a))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if a.dry_run:
        if world > 1 or a.force_collective:
            if world == 1 and "RANK" not in os.environ:
                os.environ["MASTER_PORT"] = str(_free_port())
            dist.init_process_group("gloo", rank=rank, world_size=world)   # CPU tensors: always gloo
        sys.exit(run_dry(a, rank, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback); --dry-run --backend gloo exercises the "
                         "sharded step on CPU")
    dev_index = local_rank % torch.cuda.device_count()   # one rank per GPU; wraps only in single-GPU dry runs
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if a.config == "ivfadc":
        bench_streams(a, dev)   # (before RCCL creates its streams)
    if world > 1 or a.force_collective:
        if a.config != "ivfadc":
            raise SystemExit("--config pq / join / exact are single-GPU measurements")
        if world == 1 and "RANK" not in os.environ:   # --force-collective started by hand: a one-rank group of this process
            os.environ["MASTER_PORT"] = str(_free_port())
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    if a.config == "ivfadc":
        out = run_ivfadc(a, rank, world, dev, dev_index)
        if world == 1 and not a.no_other_configs and a.N is None and a.Q is None:
            # BASELINE configs[1] and configs[3] beside the metric's configuration: bounded passes (a few seconds each)
            import copy
            other = {}
            # the reference's shipped default index shape for the same workload: K = 256 -> one byte per code (16 B per row)
            try:
                b = copy.copy(a)
                b.K, b.no_host_abi, b.no_recall, b.cpu_sample, b.steps, b.warmup = 256, True, True, 256, min(a.steps, 40), a.warmup
                torch.cuda.empty_cache()
                o = run_ivfadc(b, rank, world, dev, dev_index)
                other["ivfadc_K256"] = {"metric": o["metric"] + " -- K = 256 variant (index_creation/config/ivfadc_config.json: one byte per code)",
                                        "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
                                        "config": o["config"], "roofline": o["roofline"], "kernels": o["kernels"],
                                        "cpu_baseline": o["cpu_baseline"], "timed_region_parity": o["timed_region_parity"]}
            except Exception as e:
                other["ivfadc_K256"] = {"error": f"{type(e).__name__}: {e}"}
            for cfg, fn in (("pq", run_pq), ("join", run_join)):
                b = copy.copy(a)
                b.config, b.steps, b.warmup = cfg, min(a.steps, 20), 3
                try:
                    torch.cuda.empty_cache()
                    o = fn(b, rank, world, dev, dev_index)
                    other[cfg] = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline", "cpu_baseline",
                                                    "single_query_host_abi_ms", "single_query", "track", "kernels", "median_ms_per_step", "cached_targets", "pinned_queries", "timing") if k in o}
                except Exception as e:   # the headline line must not be lost to a side measurement
                    other[cfg] = {"error": f"{type(e).__name__}: {e}"}
            if out is not None and out.get("_exact") is not None:
                other["exact"] = out.pop("_exact")
            out["other_configs"] = other
            # the N > 1 code path on this one GPU (a child process with a one-rank RCCL group: the asynchronous all_gather,
            # option reserve_cus), with and without the collective in ONE process
            if not a.force_collective and not a.no_collective_child and a.in_flight > 1:
                try:
                    torch.cuda.empty_cache()
                    cmd = [sys.executable, os.path.abspath(__file__), "--force-collective", "--no-other-configs", "--no-host-abi", "--no-recall",
                           "--cpu-sample", "0", "--steps", str(max(a.steps, 100)), "--warmup", str(a.warmup), "--in-flight", str(a.in_flight),
                           "--backend", a.backend]
                    side = os.path.join(ROOT, "bench_details_collective.json")
                    if os.path.exists(side):
                        os.unlink(side)   # (a record of an earlier run must not stand in for a child that failed)
                    cp = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                                        env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")})
                    if cp.returncode != 0:
                        raise RuntimeError(f"child exited with {cp.returncode}: {cp.stderr.strip().splitlines()[-1] if cp.stderr.strip() else ''}")
                    det = json.load(open(side))
                    out["collective_1rank"] = det["collective_1rank"]
                    out["config"]["collective_1rank_ratio"] = det["collective_1rank"]["ratio"]
                except Exception as e:
                    out["collective_1rank"] = {"error": f"{type(e).__name__}: {e}"}
    elif a.config == "pq":
        out = run_pq(a, rank, world, dev, dev_index)
    elif a.config == "exact":
        out = run_exact(a, rank, world, dev, dev_index)
    else:
        out = run_join(a, rank, world, dev, dev_index)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out, "bench_details_collective.json" if (a.force_collective and world == 1) else "bench_details.json")


if __name__ == "__main__":
    main()
