#!/usr/bin/env python3
"""One scan launch for G guides (scan_rows_kernel loops over the guides with a tile's bit-planes staged once and register-resident):
does sharing the staging, the reverse strand's bit-reversal / complement and the dead-tile walk pay?  Scan-kernel time per launch and
per guide for G = 1, 2, 4, 8 through calitas_scan_candidates (the scan alone on the chip).
python3 tools/scan_guides.py [scale] [reps] [G ...]        (torch-free: usable under rocprofv3 --pmc)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import calitas_amd as C
from calitas_amd import synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Gs = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
spec = synth.hg38_like_spec(scale)
names, seqs = [], []
for ci, (name, length) in enumerate(spec):
    rng = np.random.default_rng([0xC3, ci])
    s = synth.random_bases(rng, length)
    if length > 100000:
        s[:10000] = ord("N"); s[-10000:] = ord("N")
        s[length // 2: length // 2 + length // 100] = ord("N")
    names.append(name); seqs.append(s)
ctx = C.Context(0)
ctx.set_reference(names, seqs)
guides = ["CTTGCCCCACAGGGCAGTAAnrg"] + synth.random_guides(0xC4, 7)
params = C.make_params(max_gaps_between_guide_and_pam=2)
lib = C._lib.lib
import ctypes
for G in Gs:
    gl = [C.Guide(g) for g in guides[:G]]
    keep = [g.to_c() for g in gl]
    arr = (C._lib.GuideT * G)(*keep)
    best = None
    for _ in range(reps):
        out, cnt = ctypes.POINTER(ctypes.c_uint32)(), ctypes.c_uint64()
        C._lib.check(ctx._h, lib.calitas_scan_candidates(ctx._h, G, arr, ctypes.byref(params), ctypes.byref(out), ctypes.byref(cnt)))
        lib.calitas_free(out)
        t = ctx.timing()["scan_kernel_ms"]
        best = t if best is None else min(best, t)
    print("G=%d: scan launch %.3f ms = %.3f ms per guide (%d records)" % (G, best, best / G, cnt.value), flush=True)
ctx.close()
