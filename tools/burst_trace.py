"""GPU-box helper: where does a SHORT burst of batches (the driver's `--steps 20 --warmup 5`) spend its time?

For every step of a burst: the host time at which its enqueue started / returned, and (HIP events on the step's stream)
when its chain began and ended on the GPU, all relative to the start of the timed region.  Run for several numbers of
batches in flight.  usage: burst_trace.py [steps] [reps] [in_flight list, e.g. 3,4,5,6,8]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("QUEUES", "6"))
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "postgres-word2vec_amd")]
import numpy as np, torch
from freddy_amd import gpu, index_build as ib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fl_list = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "4").split(",")]
verbose = os.environ.get("VERBOSE", "1") != "0"
N, Q = 3_000_000, 1024
dev = torch.device("cuda", 0)
x = ib.make_corpus(N, d=300, seed=20260101, device=dev)
tab = ib.build_ivf_index(x, C=1000, m=12, K=1024, train_size=100000, iters=10, seed=2)
index = gpu.IVFIndex(tab["coarse"], tab["codebook"], tab["list_off"], tab["ids"], tab["codes"], device=0)
rng = np.random.default_rng(7)
max_fl = max(fl_list)
d_qs = []
for _ in range(max_fl):
    qids = np.sort(rng.choice(np.arange(1, N + 1), size=Q, replace=False)).astype(np.int64)
    d_qs.append(x[torch.from_numpy(qids - 1).to(dev)].contiguous())
res = [torch.empty((2, Q, 5), dtype=torch.int32, device=dev) for _ in range(max_fl)]
st = torch.zeros(4, dtype=torch.int32, device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(max_fl)]
share_of = os.environ.get("SHARE")

for n_fl in fl_list:
    index.set_option("scan_share", int(share_of) if share_of else n_fl)
    def step(i):
        s = streams[i % n_fl]
        index.search_dev(d_qs[i % n_fl].data_ptr(), Q, 5, 10, 1000.0, gpu.FOUND_ROWS, res[i % n_fl][0].data_ptr(),
                         res[i % n_fl][1].data_ptr(), st.data_ptr(), s.cuda_stream)
    for i in range(3 * n_fl):
        step(i)
    torch.cuda.synchronize()
    totals = []
    for rep in range(reps):
        ev0 = torch.cuda.Event(enable_timing=True)
        eb = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ee = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        h0, h1 = [0.0] * steps, [0.0] * steps
        torch.cuda.synchronize()
        ev0.record(streams[0])
        t0 = time.perf_counter()
        for i in range(steps):
            s = streams[i % n_fl]
            h0[i] = time.perf_counter() - t0
            eb[i].record(s)
            step(i)
            ee[i].record(s)
            h1[i] = time.perf_counter() - t0
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        totals.append(t_all)
        if verbose and rep == reps - 1:
            print(f"--- in_flight={n_fl} steps={steps}: enqueue of all steps {1e6 * t_enq:.0f} us, done after {1e6 * t_all:.0f} us ({1e6 * t_all / steps:.1f} us per step; events add ~2 launches per step)")
            for i in range(steps):
                print(f"  step {i:2d} stream {i % n_fl}: host enqueue {1e6 * h0[i]:7.0f} .. {1e6 * h1[i]:7.0f} us | gpu chain {1e3 * ev0.elapsed_time(eb[i]):7.0f} .. {1e3 * ev0.elapsed_time(ee[i]):7.0f} us")
    # the same burst without the events (what bench.py times); EMULATE=1: the step as bench.py wrote it until round 3
    # (torch stream context + PipelinedGather bookkeeping around the call)
    emulate = os.environ.get("EMULATE") == "1"
    if emulate:
        from freddy_amd import shard
        pg = shard.PipelinedGather(Q, 5, dev, depth=max(2, n_fl))
        cnt = [0]
        def step(i):
            j = cnt[0] % n_fl
            s = streams[j]
            cnt[0] += 1
            with torch.cuda.stream(s):
                r = pg.next_buffer()
                index.search_dev(d_qs[j].data_ptr(), Q, 5, 10, 1000.0, gpu.FOUND_ROWS, r[0].data_ptr(), r[1].data_ptr(), st.data_ptr(), s.cuda_stream)
                pg.submit()
    plain = []
    for rep in range(max(reps, 5)):
        if os.environ.get("COLD") == "1":
            torch.cuda.synchronize(); time.sleep(0.3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        plain.append((time.perf_counter() - t0, t_enq))
    print("   reps (us per step):", " ".join(f"{1e6 * p[0] / steps:.1f}" for p in plain))
    if os.environ.get("RAMP") == "1":   # how long after an idle period do bursts run slow?  (clock ramp)
        for idle in (0.001, 0.01, 0.05, 0.3, 1.0):
            torch.cuda.synchronize(); time.sleep(idle)
            seq = []
            for rep in range(12):
                t0 = time.perf_counter()
                for i in range(steps):
                    step(i)
                torch.cuda.synchronize()
                seq.append(time.perf_counter() - t0)
            print(f"   after {1e3 * idle:.0f} ms idle, consecutive bursts (us per step):", " ".join(f"{1e6 * t / steps:.1f}" for t in seq))
        # busy with something else first (a big torch matmul), then one burst
        for busy_ms in (1, 5, 20, 100):
            torch.cuda.synchronize(); time.sleep(0.5)
            a_ = torch.randn(4096, 4096, device=dev); t0 = time.perf_counter()
            while time.perf_counter() - t0 < busy_ms * 1e-3:
                a_ = (a_ @ a_).clamp_(-1, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            torch.cuda.synchronize()
            print(f"   0.5 s idle, {busy_ms} ms of matmuls, then one burst: {1e6 * (time.perf_counter() - t0) / steps:.1f} us per step")
    best = min(plain)
    med = sorted(plain)[len(plain) // 2]
    print(f"in_flight={n_fl} steps={steps}: plain burst median {1e6 * med[0] / steps:.1f} us per step ({Q * steps / med[0] / 1e6:.2f} M q/s), best {1e6 * best[0] / steps:.1f}; host enqueue {1e6 * med[1] / steps:.1f} us per step", flush=True)
