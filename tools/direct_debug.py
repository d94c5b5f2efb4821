import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "postgres-word2vec_amd"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import numpy as np
import util
from freddy_amd import gpu
from oracle.oracle import Oracle
o = Oracle()
cases = [(32, 200, 1, 3, 1), (32, 64, 2, 32, 0), (3, 300, 2, 5, 0), (128, 48, 3, 5, 0), (32, 1000, 1, 5, 2)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for ci, (C, Q, W, k, rule) in enumerate(cases):
    if only >= 0 and ci != only: continue
    N = 20000
    t = util.ivf_tables(N=N, C=C, K=256)
    ot = o.ivf_table(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx = gpu.IVFIndex(t["coarse"], t["codebook"], t["list_off"], t["ids"], t["codes"])
    idx.set_option("fused", 1)
    _, qs = util.queries_from_corpus(N, Q)
    sentinel = 1000.0 if rule == 0 else 100.0
    fr = {0: gpu.FOUND_ROWS, 1: gpu.FOUND_ACCEPTED, 2: gpu.FOUND_BATCH_UDF}[rule]
    exp = o.ivfadc_batch_search(ot, qs, k) if rule == 2 else o.ivfadc_search_many(ot, qs, k, W, sentinel=sentinel, found_rule=rule)
    for direct in ((1, 0) if os.environ.get("ORDER") == "10" else (0, 1)):
        idx.set_option("direct", direct)
        print("case", ci, (C, Q, W, k, rule), "direct", direct, flush=True)
        gi, gd = idx.search(qs, k, W, sentinel=sentinel, found_rule=fr)
        ok = np.array_equal(gi, exp["id"]) and np.array_equal(gd.view(np.uint32), exp["dist"].view(np.uint32))
        print("   ->", "same" if ok else "DIFFERENT", flush=True)
    idx.close()
