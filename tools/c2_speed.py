#!/usr/bin/env python3
"""BASELINE config 2: one E. coli-sized contig (4 641 652 bp, GC 0.508, seed 0xC2), guide #0 + NRG, max-guide-diffs 3 (defaults
otherwise): time per calitas_search_hits call, and the rows against the oracle.  Usage: python3 tools/c2_speed.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import calitas_amd as C
from calitas_amd import synth
import oracle_lib as O
from fasta_util import write_fasta

GUIDE = "CTTGCCCCACAGGGCAGTAAnrg"
G = C.Guide(GUIDE)
names, seqs = synth.make_genome([("ecoli_like", 4641652)], 0xC2, guides=[(G.guide, G.pams[0], False)], sites_per_guide=40, gc=0.508, softmask=0.0,
                                n_run_ends=0, n_block=0, tandem_frac=0.0)
seq = seqs[0]
tmp = "/tmp/c2"; os.makedirs(tmp, exist_ok=True)
fa = write_fasta(tmp + "/ecoli_like.fa", [("ecoli_like", seq.tobytes().decode())])
ctx = C.Context(0)
ctx.set_reference_fasta(fa)
params = C.make_params(max_guide_diffs=3)
text, n = ctx.search_hits(G, "c2", params, "v", "t")
ts = []
for _ in range(50):
    t = time.perf_counter(); ctx.search_hits(G, "c2", params, "v", "t", decode=False); ts.append(time.perf_counter() - t)
tm = ctx.timing()
ms = 1e3 * sorted(ts)[len(ts) // 2]
print("config 2: %d rows; %.3f ms per call (median of 50), %.3g candidates/s; scan %.3f ms, align %.3f ms" % (
    n, ms, 2 * len(seq) / (ms * 1e-3), tm["scan_kernel_ms"], tm["align_kernel_ms"]))
t = time.perf_counter()
_, want, _ = O.search_reference(fa, GUIDE, "c2", d=3, threads=16)
dt = time.perf_counter() - t
SK = {"aligner_version", "time_stamp"}
same = [{k: v for k, v in r.items() if k not in SK} for r in C.read_hits(text)] == [{k: v for k, v in r.items() if k not in SK} for r in want]
print("oracle (16 threads): %.2f s, %d rows, identical: %s" % (dt, len(want), same))
ctx.close()
sys.exit(0 if same else 1)
