#!/usr/bin/env python3
"""Fresh contexts in a loop: every iteration creates a context, loads a 6.6 Mb reference and makes its first calls in a different
lane layout (and through the batch call); all texts must be identical.  First calls are where scratch is created and cleared
(the null-stream race).  Usage: python3 tools/fresh_loop.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import calitas_amd as C
from calitas_amd import synth
from fasta_util import write_fasta
guide = "CTTGCCCCACAGGGCAGTAAnrg"
G = C.Guide(guide)
names, seqs = synth.make_genome([("a", 3000000), ("b", 2000000), ("c", 1200000), ("d", 400000)], seed=77, guides=[(G.guide, G.pams[0], False)],
                                sites_per_guide=300, n_run_ends=100, n_block=1000, softmask=0.3)
params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
want = None
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for it in range(n_iter):
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="loop")
    try:
        os.environ["CALITAS_CHUNKS"] = ["3", "2", "4:3:2", "1"][it % 4]
        got = ctx.search_hits(G, "g", params, "v", "t")
        if it % 3 == 0:
            got2 = ctx.search_hits_batch([G, G, G], ["g"] * 3, params, "v", "t")
            assert got2[0] == got and got2[2] == got, "batch differs at %d" % it
        if want is None:
            want = got
        assert got == want, "iteration %d differs" % it
    finally:
        ctx.close()
print("fresh-context loop: %d iterations identical (%d rows)" % (n_iter, want[1]))
