"""What a rank of the N-GPU window partition runs (bench.py --shard windows): calitas_search_hits_into on the WHOLE resident genome with
a window range of 1/N of the windows.  ms per call (median) for every rank's range: python tools/owned_speed.py [N] [calls]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    import numpy as np
    import torch
    import bench
    import calitas_amd as C
    from calitas_amd import shard, synth
    names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
    lengths = [len(s) for s in seqs]
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="synthetic")
    del seqs
    G = C.Guide(bench.GUIDE0)
    buf = np.zeros(128 << 20, dtype=np.uint8)
    ctx.pin_host(buf.ctypes.data, buf.nbytes)
    base = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    step_w = 1000 - (len(bench.GUIDE0) + base["max_guide_diffs"] + base["max_gaps_between_guide_and_pam"] - 1)   # as bench.py's window mode
    parts = shard.window_partition(lengths, n, step_w)
    worst = 0.0
    for r, (first, cnt) in enumerate(parts):
        p = C.make_params(first_window=first, n_windows=cnt, **base)
        times = []
        for i in range(calls + 5):
            t0 = time.perf_counter()
            nb, rows = ctx.search_hits_into(G, "a", p, buf.ctypes.data, buf.nbytes, "v0", "stamp")
            if i >= 5:
                times.append((time.perf_counter() - t0) * 1e3)
        times.sort()
        tm = ctx.timing()
        worst = max(worst, times[len(times) // 2])
        print("rank %d of %d: windows %d + %d  median %.3f ms  min %.3f  lanes %d binned %d  scan %.3f align+trace %.3f rows %.3f copy %.3f  rows %d bytes %d" % (
            r, n, first, cnt, times[len(times) // 2], times[0], tm["lanes"], tm["binned_lanes"], tm["scan_kernel_ms"], tm["align_kernel_ms"], tm["hits_kernel_ms"],
            tm["hits_copy_ms"], rows, nb), flush=True)
    print("slowest rank: %.3f ms" % worst)
    ctx.unpin_host(buf.ctypes.data)
    ctx.close()


if __name__ == "__main__":
    main()
