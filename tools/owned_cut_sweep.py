"""Range cuts of a rank's call (a window range of 1/N of the genome), interleaved call by call in one process:
python tools/owned_cut_sweep.py CALLS CUT1 CUT2 ...   ("-" = the default); ranks 0 and N-2 of N = 8, 4, 2."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
calls = int(sys.argv[1])
cuts = sys.argv[2:]
import numpy as np
import torch
import bench
import calitas_amd as C
from calitas_amd import shard
names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
lengths = [len(s) for s in seqs]
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="synthetic")
del seqs
G = C.Guide(bench.GUIDE0)
buf = np.zeros(256 << 20, dtype=np.uint8)
ctx.pin_host(buf.ctypes.data, buf.nbytes)
base = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
step_w = 1000 - (len(bench.GUIDE0) + base["max_guide_diffs"] + base["max_gaps_between_guide_and_pam"] - 1)
for n in (8, 4, 2):
    parts = shard.window_partition(lengths, n, step_w)
    for rank in sorted({0, n - 2} if n > 2 else {0, 1}):
        first, cnt = parts[rank]
        p = C.make_params(first_window=first, n_windows=cnt, **base)
        times = {c: [] for c in cuts}
        for i in range(calls + 4):
            for c in cuts:
                if c == "-":
                    os.environ.pop("CALITAS_CHUNKS", None)
                else:
                    os.environ["CALITAS_CHUNKS"] = c
                t0 = time.perf_counter()
                ctx.search_hits_into(G, "a", p, buf.ctypes.data, buf.nbytes, "v0", "stamp")
                if i >= 4:
                    times[c].append((time.perf_counter() - t0) * 1e3)
        print("rank %d of %d: " % (rank, n) + "   ".join("%s %.3f" % (c, sorted(times[c])[len(times[c]) // 2]) for c in cuts), flush=True)
ctx.close()
