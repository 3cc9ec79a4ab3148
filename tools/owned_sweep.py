"""A rank's window range (rank R of N) under several values of one environment switch, interleaved call by call in one process:
python tools/owned_sweep.py VAR R N CALLS V1 V2 ...   ("-" = unset)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    var, r, n, calls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    values = sys.argv[5:]
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
    step_w = 1000 - (len(bench.GUIDE0) + 5 + 2 - 1)
    first, cnt = shard.window_partition(lengths, n, step_w)[r]
    p = C.make_params(first_window=first, n_windows=cnt, **base)
    res = {v: [] for v in values}
    sums = {v: [0.0, 0.0, 0.0] for v in values}
    for i in range(len(values) * (calls + 5)):
        v = values[i % len(values)]
        if v == "-":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        t0 = time.perf_counter()
        ctx.search_hits_into(G, "a", p, buf.ctypes.data, buf.nbytes, "v0", "stamp")
        dt = (time.perf_counter() - t0) * 1e3
        if i >= 5 * len(values):
            res[v].append(dt)
            tm = ctx.timing()
            for k, key in enumerate(("scan_kernel_ms", "align_kernel_ms", "hits_kernel_ms")):
                sums[v][k] += tm[key]
    for v in values:
        t = sorted(res[v])
        print("%s=%-6s rank %d of %d: median %.3f ms  min %.3f  p75 %.3f | scan %.3f align+trace %.3f rows %.3f" % (
            var, v, r, n, t[len(t) // 2], t[0], t[3 * len(t) // 4], sums[v][0] / len(t), sums[v][1] / len(t), sums[v][2] / len(t)), flush=True)
    ctx.unpin_host(buf.ctypes.data)
    ctx.close()


if __name__ == "__main__":
    main()
