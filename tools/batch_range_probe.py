"""calitas_search_hits_batch on a rank's window range (BASELINE config 4 sharded over N GPUs): how many of the 96 guides the per-bin
kernels decide, and what a step costs.  python tools/batch_range_probe.py [scale] [ranks] [rank] [guides]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import calitas_amd as C
from calitas_amd import shard, synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_g = int(sys.argv[4]) if len(sys.argv) > 4 else 8
names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="x")
lengths = [len(s) for s in seqs]
kw = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
first, n = shard.window_partition(lengths, ranks, 971)[rank]
pr = C.make_params(first_window=first, n_windows=n, **kw)
guides = [bench.GUIDE0] + synth.random_guides(0xC4, n_g - 1)
G = [C.Guide(g) for g in guides]
ids = ["g%d" % i for i in range(len(G))]
for it in range(3):
    t = time.perf_counter()
    res = ctx.search_hits_batch(G, ids, pr, "v", "t", decode=False)
    tm = ctx.timing()
    print("batch of %d guides on rank %d/%d's range: %.2f ms, %d guides on the per-bin kernels, %d rows, %.1f MB" % (
        len(G), rank, ranks, (time.perf_counter() - t) * 1e3, tm["binned_lanes"], sum(r for _, r in res), sum(b for b, _ in res) / 1e6), flush=True)
ctx.close()
