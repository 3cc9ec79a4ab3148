"""Host-side time lines (CALITAS_TRACE=2) of the slowest calitas_search_hits calls of a run: python tools/slow_calls.py [calls] [scale]
Prints the median call time and, for the calls beyond 1.3x the median, the marks of all their lanes."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
import torch
import bench
import calitas_amd as C
names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="x")
del seqs
G = C.Guide(bench.GUIDE0)
params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
for _ in range(20):
    ctx.search_hits(G, "a", params, "v", "t", decode=False)
os.environ["CALITAS_TRACE"] = "2"
log = tempfile.TemporaryFile()
saved = os.dup(2)
os.dup2(log.fileno(), 2)
times = []
try:
    for i in range(calls):
        os.write(2, b"=== call %d\n" % i)
        t = time.perf_counter()
        ctx.search_hits(G, "a", params, "v", "t", decode=False)
        times.append((time.perf_counter() - t) * 1e3)
finally:
    os.dup2(saved, 2)
log.seek(0)
blocks = log.read().decode(errors="replace").split("=== call ")[1:]
med = sorted(times)[len(times) // 2]
print("median %.3f ms, mean %.3f, max %.3f, calls beyond 1.3x the median: %d of %d" % (med, sum(times) / len(times), max(times), sum(1 for t in times if t > 1.3 * med), len(times)))
shown = 0
for i, t in enumerate(times):
    if t > 1.3 * med and shown < 6:
        shown += 1
        print("--- call %d: %.3f ms" % (i, t))
        print("\n".join(ln for ln in blocks[i].splitlines()[1:] if "host marks" in ln))
ctx.close()
