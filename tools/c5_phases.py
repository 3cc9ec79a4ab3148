"""Where a config-5-shaped calitas_search_hits call spends its wall time outside the device stages: the call, then calitas_free.
python tools/c5_phases.py [scale]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import calitas_amd as C
from calitas_amd import _lib
lib = _lib.lib
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="synthetic")
g = C.Guide("GTGACTTGAAGTCTCAGTAT").to_c()
p = C.make_params(max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
for it in range(2):
    tsv, nbytes, rows = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
    t0 = time.perf_counter()
    rc = lib.calitas_search_hits(ctx._h, ctypes.byref(g), b"c5", ctypes.byref(p), b"v", b"t", ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(rows))
    t1 = time.perf_counter()
    lib.calitas_free(tsv)
    t2 = time.perf_counter()
    print("rc %d: call %.1f ms, free %.1f ms, %d rows, %d bytes" % (rc, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rows.value, nbytes.value), flush=True)
ctx.close()
