"""calitas_search_hits on the bench genome under different lane cuts (CALITAS_CHUNKS): ms per call, median of `calls`.
python tools/sweep_lanes.py [scale] [calls] [cut ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    cuts = sys.argv[3:] or ["5:3:2", "1", "4:3:2:1", "6:3:1", "5:3:1.5:0.5", "3:3:2:1.5:0.5", "5:3:2:1:0.5"]
    import torch
    import bench
    import calitas_amd as C
    names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="synthetic")
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = C.Guide(bench.GUIDE0)
    import numpy as np
    buf = np.zeros(256 << 20, dtype=np.uint8)          # the caller's text buffer, page-locked as bench.py does
    ctx.pin_host(buf.ctypes.data, buf.nbytes)
    for cut in cuts:
        os.environ["CALITAS_CHUNKS"] = cut
        times = []
        for i in range(calls + 3):
            t0 = time.perf_counter()
            ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")
            if i >= 3:
                times.append((time.perf_counter() - t0) * 1e3)
        times.sort()
        print("%-16s median %.3f ms  min %.3f ms  lanes %d" % (cut, times[len(times) // 2], times[0], ctx.timing()["lanes"]), flush=True)
    ctx.unpin_host(buf.ctypes.data)
    ctx.close()


if __name__ == "__main__":
    main()
