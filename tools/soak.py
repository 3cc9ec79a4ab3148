#!/usr/bin/env python3
"""Soak: many guides at full (hg38-sized) scale; the one-pass, lanes, batch and two-stage paths must return identical bytes for
every guide, call after call.  Usage: python3 tools/soak.py [n_guides] [scale]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import calitas_amd as C
from calitas_amd import synth

n_guides = int(sys.argv[1]) if len(sys.argv) > 1 else 12
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="soak")
guides = [bench.GUIDE0] + synth.random_guides(0x50AC, n_guides - 1) + ["tttvAACCAACCAACCGGTTACGT", "GTGACTTGAAGTCTCAGTATA"]
params = C.make_params(max_guide_diffs=4, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
bad = 0
t0 = time.time()
for gi, g in enumerate(guides):
    G = C.Guide(g)
    os.environ.pop("CALITAS_CHUNKS", None)
    a = ctx.search_hits(G, "g", params, "v", "t")
    os.environ["CALITAS_CHUNKS"] = "1"
    b = ctx.search_hits(G, "g", params, "v", "t")
    os.environ["CALITAS_CHUNKS"] = "3"
    c = ctx.search_hits(G, "g", params, "v", "t")
    os.environ.pop("CALITAS_CHUNKS")
    out, k = ctx.search_raw([G], params)
    d = ctx.hits_tsv_raw(G, "g", params, out, k, "v", "t")
    C._lib.lib.calitas_free(out)
    again = ctx.search_hits(G, "g", params, "v", "t")
    ok = a == b == c == d == again
    bad += not ok
    print("guide %2d %-26s rows %7d  %s" % (gi, g, a[1], "ok" if ok else "MISMATCH %s" % [x[1] for x in (a, b, c, d, again)]), flush=True)
same_len = [g for g in guides if len(g) == len(guides[0])]
batch = ctx.search_hits_batch([C.Guide(g) for g in same_len], ["g"] * len(same_len), params, "v", "t")
singles = [ctx.search_hits(C.Guide(g), "g", params, "v", "t") for g in same_len]
print("batch of %d: %s" % (len(same_len), "ok" if batch == singles else "MISMATCH"), flush=True)
bad += batch != singles
print("soak: %d guides, %d mismatches, %.1f s" % (len(guides), bad, time.time() - t0))
ctx.close()
sys.exit(1 if bad else 0)
