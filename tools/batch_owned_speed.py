"""What every rank of `bench.py --gpus N`'s batch_sharded block runs, timed for all ranks of N on one MI355X: calitas_search_hits_batch with
the 96 guides of BASELINE config 4 on the rank's window range of the hg38-sized genome.  The slowest rank is the job's step; N = 1 is
the whole genome.  python tools/batch_owned_speed.py [scale] [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import calitas_amd as C
from calitas_amd import shard, synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Ns = sorted([int(x) for x in sys.argv[2:]] or [1, 2, 4, 8], reverse=True)   # small texts first: the pool of page-locked blocks keeps the big ones
names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="x")
lengths = [len(s) for s in seqs]
kw = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
G = [C.Guide(g) for g in [bench.GUIDE0] + synth.random_guides(0xC4, 95)]
ids = ["g%02d" % i for i in range(len(G))]
results = {}
for N in Ns:
    worst, detail = 0.0, []
    for rank in range(N):
        if N == 1:
            pr = C.make_params(**kw)
        else:
            first, n = shard.window_partition(lengths, N, 971)[rank]
            pr = C.make_params(first_window=first, n_windows=n, **kw)
        ctx.search_hits_batch(G, ids, pr, "v", "t", decode=False)
        best = 1e9
        for it in range(2):
            t = time.perf_counter()
            res = ctx.search_hits_batch(G, ids, pr, "v", "t", decode=False)
            best = min(best, (time.perf_counter() - t) * 1e3)
        tm = ctx.timing()
        detail.append("%.1f(%d+%d)" % (best, tm["binned_lanes"], tm["owned_general_lanes"]))
        worst = max(worst, best)
    results[N] = (worst, detail)
ctx.close()
base = results.get(1, (None,))[0]
for N in sorted(results):
    worst, detail = results[N]
    print("N=%d: slowest rank %.1f ms per 96-guide step%s  ranks ms(guides on the bins + finished by the general kernels): %s"
          % (N, worst, "  (%.2fx of N=1)" % (base / worst) if base else "", " ".join(detail)), flush=True)
