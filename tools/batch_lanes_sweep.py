import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
import calitas_amd as C
from calitas_amd import synth
names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0); ctx.set_reference(names, seqs, genome_build="x"); del seqs
guides = ([bench.GUIDE0] + synth.random_guides(0xC4, 95))
G = [C.Guide(g) for g in guides]; ids = ["g%02d" % i for i in range(96)]
p = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
ctx.search_hits_batch(G, ids, p, "v", "t", decode=False)
res = {}
for rep in range(int(os.environ.get("SWEEP_REPS", "4"))):
    for lanes in sys.argv[2:] or ("4", "5", "6", "7", "8"):
        var = sys.argv[1] if len(sys.argv) > 1 else "CALITAS_BATCH_LANES"
        if lanes == "-":
            os.environ.pop(var, None)                     # ("-": the variable unset)
        else:
            os.environ[var] = lanes
        t = time.perf_counter(); r = ctx.search_hits_batch(G, ids, p, "v", "t", decode=False); dt = (time.perf_counter() - t) * 1e3; del r
        res.setdefault(lanes, []).append(dt)
for k, v in res.items(): print("%s=%s: %s ms per 96 guides" % (sys.argv[1] if len(sys.argv) > 1 else "CALITAS_BATCH_LANES", k, " ".join("%.1f" % x for x in v)), flush=True)
ctx.close()
