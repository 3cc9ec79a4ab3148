"""Several values of one environment switch of the library on calitas_search_hits, interleaved call by call in ONE process on one box
(like tools/ab_env.py, for more than two settings): python tools/sweep_env.py VAR SCALE CALLS V1 V2 V3 ... [-- VAR2=VALUE ...]
"-" leaves the variable unset.  Prints median / min of the call time and the mean per-lane kernel sums per value."""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    var, scale, calls = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    rest = sys.argv[4:]
    extra = []
    if "--" in rest:
        extra = rest[rest.index("--") + 1:]
        rest = rest[:rest.index("--")]
    for kv in extra:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    values = rest
    import numpy as np
    import torch
    import bench
    import calitas_amd as C
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = C.Guide(bench.GUIDE0)
    buf = np.zeros(256 << 20, dtype=np.uint8)
    names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="synthetic")
    del seqs
    ctx.pin_host(buf.ctypes.data, buf.nbytes)
    res = {v: [] for v in values}
    sums = {v: [0.0] * 4 for v in values}
    crc = {}
    t_end = time.perf_counter() + 0.75
    while time.perf_counter() < t_end:
        ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")
    n_v = len(values)
    for i in range(n_v * (calls + 4)):
        v = values[i % n_v]
        if v == "-":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        t0 = time.perf_counter()
        n_bytes, _ = ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")[:2]
        dt = (time.perf_counter() - t0) * 1e3
        if i >= 4 * n_v:
            res[v].append(dt)
            tm = ctx.timing()
            for k, key in enumerate(("scan_kernel_ms", "align_kernel_ms", "hits_kernel_ms", "hits_copy_ms")):
                sums[v][k] += tm[key]
        if v not in crc:
            crc[v] = zlib.crc32(bytes(buf[:int(n_bytes)]))
    for v in values:
        t = sorted(res[v])
        n = len(t)
        print("%s=%-10s scale %g: median %.3f ms  mean %.3f  min %.3f  p25 %.3f  p75 %.3f  p95 %.3f  max %.3f | scan %.3f align+trace %.3f rows %.3f copy %.3f (mean sums over lanes, ms)  crc %08x" % (
            var, v, scale, t[n // 2], sum(t) / n, t[0], t[n // 4], t[3 * n // 4], t[min(n - 1, n * 95 // 100)], t[-1], sums[v][0] / n, sums[v][1] / n, sums[v][2] / n, sums[v][3] / n, crc[v]), flush=True)
    ctx.unpin_host(buf.ctypes.data)
    ctx.close()


if __name__ == "__main__":
    main()
