"""CPU time the process group burns per calitas_search_hits call on the bench genome, and how often the box throttled it meanwhile
(cgroup cpu.stat): python tools/cpu_use.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
import torch
import bench
import calitas_amd as C

def stat():
    out = {}
    d = "/sys/fs/cgroup" + open("/proc/self/cgroup").read().split(":")[-1].strip()
    while True:
        try:
            for ln in open(d + "/cpu.stat"):
                k, v = ln.split()
                out.setdefault(k, int(v)) if k in ("nr_throttled", "throttled_usec", "nr_periods") and int(v) else None
            if "usage_usec" not in out:
                for ln in open(d + "/cpu.stat"):
                    k, v = ln.split()
                    if k == "usage_usec":
                        out[k] = int(v)
        except OSError:
            pass
        if d in ("/sys/fs/cgroup", "/", ""):
            break
        d = os.path.dirname(d)
    return out

names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="x")
del seqs
G = C.Guide(bench.GUIDE0)
params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
for _ in range(20):
    ctx.search_hits(G, "a", params, "v", "t", decode=False)
time.sleep(0.5)
s0, t0, c0 = stat(), time.perf_counter(), time.process_time()
times = []
for _ in range(calls):
    t = time.perf_counter()
    ctx.search_hits(G, "a", params, "v", "t", decode=False)
    times.append(time.perf_counter() - t)
s1, t1, c1 = stat(), time.perf_counter(), time.process_time()
wall = t1 - t0
print("calls %d, wall %.3f s, median %.3f ms, mean %.3f ms" % (calls, wall, sorted(times)[calls // 2] * 1e3, sum(times) / calls * 1e3))
print("process CPU time %.3f s = %.2f cores busy; cgroup usage %.3f s = %.2f cores" % (c1 - c0, (c1 - c0) / wall, (s1.get("usage_usec", 0) - s0.get("usage_usec", 0)) / 1e6, (s1.get("usage_usec", 0) - s0.get("usage_usec", 0)) / 1e6 / wall))
print("throttled %d times, %.1f ms, in %d periods" % (s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3, s1.get("nr_periods", 0) - s0.get("nr_periods", 0)))
ctx.close()
