"""How much the tails cost the scans of a chunked call: the same hg38-sized call at max-guide-diffs 5 (1.2e5 scan records, 1.1e5 rows) and
at max-guide-diffs 1 (a few hundred records: the tails are empty, the scan does the same work).  python tools/scan_interference.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import calitas_amd as C
names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0); ctx.set_reference(names, seqs, genome_build="synthetic"); del seqs
G = C.Guide(bench.GUIDE0)
buf = np.zeros(256 << 20, dtype=np.uint8); ctx.pin_host(buf.ctypes.data, buf.nbytes)
for d in (5, 1, 5, 1):
    params = C.make_params(max_guide_diffs=d, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    ts = []
    for i in range(25):
        t0 = time.perf_counter(); ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")
        if i >= 5: ts.append((time.perf_counter() - t0) * 1e3)
    tm = ctx.timing(); ts.sort()
    print("d=%d: median %.3f ms; scan kernels (sum) %.3f ms, align+trace %.3f, rows kernel %.3f, copy %.3f; records %d rows %d" % (
        d, ts[len(ts) // 2], tm["scan_kernel_ms"], tm["align_kernel_ms"], tm["hits_kernel_ms"], tm["hits_copy_ms"], tm["scan_records"], tm["hit_rows"]), flush=True)
ctx.unpin_host(buf.ctypes.data); ctx.close()
