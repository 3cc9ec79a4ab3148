"""One rank of the N-GPU window partition, a few calls (a driver for tools/timeline.sh: TIMELINE_PROG=tools/owned_one.py; host marks with
CALITAS_TRACE=2 on the last call): python tools/owned_one.py [N] [rank] [calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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
buf = np.zeros(128 << 20, dtype=np.uint8)
ctx.pin_host(buf.ctypes.data, buf.nbytes)
base = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
step_w = 1000 - (len(bench.GUIDE0) + base["max_guide_diffs"] + base["max_gaps_between_guide_and_pam"] - 1)
first, cnt = shard.window_partition(lengths, n, step_w)[rank]
p = C.make_params(first_window=first, n_windows=cnt, **base)
for i in range(calls):
    if i == calls - 1 and os.environ.get("OWNED_ONE_MARKS"):
        os.environ["CALITAS_TRACE"] = "2"
    nb, rows = ctx.search_hits_into(G, "a", p, buf.ctypes.data, buf.nbytes, "v0", "stamp")
print("rank %d of %d: %d rows, %d bytes" % (rank, n, rows, nb))
ctx.close()
