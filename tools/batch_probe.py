"""Where a 96-guide batch (BASELINE config 4) spends the chip: the batch as it is, and the same scans with next to nothing behind them
(max-guide-diffs 1: a few hundred hits per guide), per guide.  python tools/batch_probe.py [guides] [repeats]
Environment switches (CALITAS_BATCH_LANES ...) apply as set by the caller."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n_guides = int(sys.argv[1]) if len(sys.argv) > 1 else 96
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
import torch
import bench
import calitas_amd as C
from calitas_amd import synth
names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="x")
del seqs
guides = ([bench.GUIDE0] + synth.random_guides(0xC4, 95))[:n_guides]
G = [C.Guide(g) for g in guides]
ids = ["g%02d" % i for i in range(len(G))]
for label, kw in (("d=5 (config 4)", dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)),
                  ("d=1 (scans with almost no tail)", dict(max_guide_diffs=1, max_pam_mismatches=0, max_gaps_between_guide_and_pam=0))):
    p = C.make_params(**kw)
    ctx.search_hits_batch(G, ids, p, "v", "t", decode=False)
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        res = ctx.search_hits_batch(G, ids, p, "v", "t", decode=False)
        dt = (time.perf_counter() - t) * 1e3
        best = dt if best is None else min(best, dt)
    tm = ctx.timing()
    print("%-34s %8.1f ms per %d guides = %.3f ms per guide; rows %d bytes %.1f MB; scan kernel sum %.1f ms (%.3f per guide) align %.1f" % (
        label, best, len(G), best / len(G), sum(r for _, r in res), sum(b for b, _ in res) / 1e6, tm["scan_kernel_ms"], tm["scan_kernel_ms"] / len(G), tm["align_kernel_ms"]))
ctx.close()
