"""Codegen guard (CPU suite; hipcc cross-compiles gfx950 without a GPU).

The hot kernels are tuned to the register budget of their occupancy -- ivf_filter5_kernel runs sixteen waves per CU
(128 registers per lane), the one-wave merge lives in the CUs the scans leave -- and this compiler's register
allocation around them has moved with small source changes before (profiles/HISTORY.md: "The removed switches were
load-bearing").  A ROCm bump or a clean-up that pushes one of them over its budget would cost several per cent of the
headline without failing any parity test.  This test compiles ONE instantiation of each kernel from its header (the way
tools/lab/probe_scan.hip does: a few seconds each, in parallel), reads hipcc's -Rpass-analysis=kernel-resource-usage
remarks and fails when registers, spills or scratch exceed the ceilings committed in tests/golden/codegen_ceilings.json.
Lower figures pass (and should then be committed as the new ceilings)."""
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "postgres-word2vec_amd", "csrc")
CEILINGS = os.path.join(ROOT, "tests", "golden", "codegen_ceilings.json")

# name -> (header, the instantiation the product launches on the path named in DESIGN.md, a substring of its mangled name)
PROBES = {
    "ivf_filter5_kernel<12,FULLK>": ("fused5.h", "ivf_filter5_kernel<12, true, false>", "ivf_filter5_kernelILi12ELb1ELb0ELb0ELb0EE"),
    "ivf_filter5_kernel<12,U8>": ("fused5.h", "ivf_filter5_kernel<12, false, false, false, true>", "ivf_filter5_kernelILi12ELb0ELb0ELb0ELb1EE"),
    "ivf_filter8_kernel<12>": ("fused8.h", "ivf_filter8_kernel<12, false>", "ivf_filter8_kernelILi12ELb0ELb0EE"),
    "merge_refine_kernel<25,12,1>": ("fused5.h", "merge_refine_kernel<25, 12, 1>", "merge_refine_kernelILi25ELi12ELi1ELb0ELb0EE"),
    "merge_refine_kernel<25,12,4>": ("fused5.h", "merge_refine_kernel<25, 12, 4>", "merge_refine_kernelILi25ELi12ELi4ELb0ELb0EE"),
    "exf_filter_kernel<2,false>": ("exact2.h", "exf_filter_kernel<2, false>", "exf_filter_kernelILi2ELb0EE"),
    "join_query_kernel<1>": ("join.h", "join_query_kernel<1>", "join_query_kernelILi1ELb0EE"),
}
FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes", "SGPRs Spill": "sgpr_spill",
          "VGPRs Spill": "vgpr_spill", "Occupancy [waves/SIMD]": "occupancy"}


def hipcc_flags():
    import __graft_entry__ as g
    return [f for f in g.HIPCC_FLAGS if f != "-fPIC"]


def probe(name, tmp):
    header, inst, mangled = PROBES[name]
    src = os.path.join(tmp, re.sub(r"\W+", "_", name) + ".hip")
    with open(src, "w") as f:
        f.write(f'#include "{header}"\nusing namespace freddy;\nconst void* probe_kernels[] = {{(const void*)&{inst}}};\n')
    cmd = [os.environ.get("HIPCC", "hipcc")] + hipcc_flags() + ["-I" + CSRC, "-c", "-o", src[:-4] + ".o", src,
                                                                 "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    got, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            continue
        if cur is None or mangled not in cur:
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and m.group(1).strip() in FIELDS:
            got[FIELDS[m.group(1).strip()]] = int(m.group(2))
    assert set(got) == set(FIELDS.values()), f"{name}: resource remarks not found (got {got})"
    return got


def measure(tmp):
    with ThreadPoolExecutor(max_workers=min(7, os.cpu_count() or 1)) as ex:
        return dict(zip(PROBES, ex.map(lambda n: probe(n, tmp), PROBES)))


@pytest.mark.skipif(shutil.which(os.environ.get("HIPCC", "hipcc")) is None, reason="hipcc not on PATH")
def test_hot_kernels_stay_within_their_register_budget(tmp_path):
    ceilings = json.load(open(CEILINGS))
    got = measure(str(tmp_path))
    bad = []
    for name, g in got.items():
        c = ceilings[name]
        for k in ("vgprs", "agprs", "scratch_bytes", "sgpr_spill", "vgpr_spill"):
            if g[k] > c[k]:
                bad.append(f"{name}: {k} = {g[k]} > ceiling {c[k]}")
        if g["occupancy"] < c["occupancy"]:
            bad.append(f"{name}: occupancy = {g['occupancy']} waves/SIMD < {c['occupancy']}")
    assert not bad, "\n".join(bad) + "\n(measured: " + json.dumps(got) + ")"


if __name__ == "__main__":   # python tests/test_codegen.py [--write]: print (and commit) today's figures
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        res = measure(td)
    print(json.dumps(res, indent=1))
    if "--write" in sys.argv:
        with open(CEILINGS, "w") as f:
            json.dump(res, f, indent=1)
            f.write("\n")
