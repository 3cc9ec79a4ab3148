"""A/B of one environment switch of the library on calitas_search_hits, interleaved call by call in ONE process on one box (two
separate bench runs differ by +-0.04 ms from box noise alone): python tools/ab_env.py VAR A B [scale] [calls] [VAR2=VALUE ...]
A value of "-" leaves the variable unset.  Prints the median / min of the call time and the mean per-lane kernel sums per setting."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
    scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    calls = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    for kv in sys.argv[6:]:
        k, v = kv.split("=", 1)
        os.environ[k] = v
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
    res = {va: [], vb: []}
    sums = {va: [0.0] * 4, vb: [0.0] * 4}
    texts = {}
    t_end = time.perf_counter() + 0.75
    while time.perf_counter() < t_end:                      # sustained clocks first
        ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")
    for i in range(2 * calls + 8):
        v = va if i % 2 == 0 else vb
        if v == "-":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        t0 = time.perf_counter()
        n_bytes, _ = ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")[:2]
        dt = (time.perf_counter() - t0) * 1e3
        if i >= 8:
            res[v].append(dt)
            tm = ctx.timing()
            for k, key in enumerate(("scan_kernel_ms", "align_kernel_ms", "hits_kernel_ms", "hits_copy_ms")):
                sums[v][k] += tm[key]
        if v not in texts:
            import zlib
            texts[v] = zlib.crc32(bytes(buf[:int(n_bytes)]))
    for v in (va, vb):
        t = sorted(res[v])
        n = len(t)
        print("%s=%-6s scale %g: median %.3f ms  min %.3f  p25 %.3f  p75 %.3f | scan %.3f align+trace %.3f rows %.3f copy %.3f (mean sums over lanes, ms)  crc %08x" % (
            var, v, scale, t[n // 2], t[0], t[n // 4], t[3 * n // 4], sums[v][0] / n, sums[v][1] / n, sums[v][2] / n, sums[v][3] / n, texts[v]), flush=True)
    ctx.unpin_host(buf.ctypes.data)
    ctx.close()


if __name__ == "__main__":
    main()
