"""What binds a 96-guide batch (BASELINE config 4): its kernels, or its texts' way home?  Needs the experiments build of the library
(make -C calitas_amd/csrc EXPERIMENTS=1; CALITAS_LIB_PATH=calitas_amd/libcalitas_hip_exp.so): the same batch, interleaved in one
process, (a) as shipped, (b) CALITAS_BATCH_TEXT=copy -- the compact rows cross the bus, nobody expands them --, (c) =skip -- the rows
stay on the device.  (b) and (c) return wrong texts: timing only.  python tools/batch_text_probe.py [guides] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n_guides = int(sys.argv[1]) if len(sys.argv) > 1 else 96
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
p = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
ctx.search_hits_batch(G, ids, p, "v", "t", decode=False)
modes = ("-", "copy", "skip")
times = {m: [] for m in modes}
info = {}
for r in range(reps):
    for m in modes:
        if m == "-":
            os.environ.pop("CALITAS_BATCH_TEXT", None)
        else:
            os.environ["CALITAS_BATCH_TEXT"] = m
        t = time.perf_counter()
        res = ctx.search_hits_batch(G, ids, p, "v", "t", decode=False)
        times[m].append((time.perf_counter() - t) * 1e3)
        tm = ctx.timing()
        info[m] = (sum(r_ for _, r_ in res), sum(b for b, _ in res), tm["scan_kernel_ms"], tm["align_kernel_ms"])
        del res
os.environ.pop("CALITAS_BATCH_TEXT", None)
for m in modes:
    rows, nbytes, scan, align = info[m]
    print("CALITAS_BATCH_TEXT=%-5s %s ms per %d guides (best %.3f per guide); rows %d, bytes handed over %.1f MB; scan kernels %.1f ms (%.3f per guide), align %.1f" % (
        m, " ".join("%.1f" % x for x in times[m]), len(G), min(times[m]) / len(G), rows, nbytes / 1e6, scan, scan / len(G), align), flush=True)
ctx.close()
