#!/usr/bin/env python3
"""Torch-free driver for rocprofv3 counter passes: builds an hg38-like synthetic genome with numpy and runs K
SearchReference passes through the C ABI.  Usage: python3 tools/scan_profile.py [scale] [steps] [hits]
(with a third argument the passes go through calitas_search_hits, so the filter / hits / row kernels run as well)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import calitas_amd as C
from calitas_amd import synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
spec = synth.hg38_like_spec(scale)
names, seqs = [], []
t0 = time.time()
for ci, (name, length) in enumerate(spec):
    rng = np.random.default_rng([0xC3, ci])
    s = synth.random_bases(rng, length)
    if length > 100000:
        s[:10000] = ord("N"); s[-10000:] = ord("N")
        s[length // 2: length // 2 + length // 100] = ord("N")
    names.append(name); seqs.append(s)
print("genome %d bp in %.1f s" % (sum(len(s) for s in seqs), time.time() - t0), flush=True)
ctx = C.Context(0)
ctx.set_reference(names, seqs)
G = [C.Guide("CTTGCCCCACAGGGCAGTAAnrg")]
params = C.make_params(max_gaps_between_guide_and_pam=2)
ts = []
fused = len(sys.argv) > 3
for i in range(steps):
    if fused:
        _, n = ctx.search_hits(G[0], "p", params, "v", "t", decode=False)
    else:
        out, n = ctx.search_raw(G, params)
        C._lib.lib.calitas_free(out)
    t = ctx.timing()
    ts.append(t)
    print("step %d: scan %.3f ms align %.3f ms, %d alignments, packed bytes %d" % (i, t["scan_kernel_ms"], t["align_kernel_ms"], n, t["packed_bytes"]), flush=True)
print("min over %d steps: scan %.3f ms, align %.3f ms, gpu_total %.3f ms, host filter %.3f ms" % (
    steps, min(t["scan_kernel_ms"] for t in ts), min(t["align_kernel_ms"] for t in ts), min(t["gpu_total_ms"] for t in ts),
    min(t["host_post_ms"] for t in ts)), flush=True)
ctx.close()
