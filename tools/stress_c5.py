#!/usr/bin/env python3
"""BASELINE config-5 shape without the VCF: PAM-less 20-mer, max-guide-diffs 8, on a synthetic genome of the given scale.
Usage: python3 tools/stress_c5.py SCALE [d] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import calitas_amd as C

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
names, seqs = bench.build_genome(scale, dev, contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="synthetic")
g = C.Guide("GTGACTTGAAGTCTCAGTAT")
p = C.make_params(max_guide_diffs=d, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    t = time.perf_counter()
    try:
        if os.environ.get("STRESS_STREAM"):          # calitas_search_hits_stream: pieces are counted and dropped
            got = [0]
            def take(piece):
                got[0] += len(piece)
            nbytes, rows = ctx.search_hits_stream(g, "c5", p, take, "v", "t")
            assert got[0] == nbytes
        else:
            nbytes, rows = ctx.search_hits(g, "c5", p, "v", "t", decode=False)
    except Exception as e:
        print("FAILED after %.1f ms: %s" % ((time.perf_counter() - t) * 1e3, e)); break
    tm = ctx.timing()
    print("scale %.3f d=%d: %.1f ms, %d rows, %.1f MB text, records %d raw %d accepted %d retries %d lanes %d scan %.2f align %.2f hits %.2f copy %.2f" % (
        scale, d, (time.perf_counter() - t) * 1e3, rows, nbytes / 1e6, tm["scan_records"], tm["raw_alignments"], tm["accepted_alignments"],
        tm["retries"], tm["lanes"], tm["scan_kernel_ms"], tm["align_kernel_ms"], tm["hits_kernel_ms"], tm["hits_copy_ms"]), flush=True)
ctx.close()
