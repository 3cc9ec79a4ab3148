"""calitas_search_hits on slices of the bench genome (a rank's share at 1 / 2 / 4 / 8 GPUs): ms per call (median), with the per-bin
tail (binned.hip, default) and with the general kernels (CALITAS_BINNED=0).  python tools/slice_speed.py [calls] [scale ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    scales = [float(x) for x in sys.argv[2:]] or [1.0, 0.5, 0.25, 0.125]
    import numpy as np
    import torch
    import bench
    import calitas_amd as C
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = C.Guide(bench.GUIDE0)
    buf = np.zeros(256 << 20, dtype=np.uint8)
    for scale in scales:
        names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
        ctx = C.Context(0)
        ctx.set_reference(names, seqs, genome_build="synthetic")
        del seqs
        ctx.pin_host(buf.ctypes.data, buf.nbytes)
        for mode in ("binned", "general"):
            if mode == "general":
                os.environ["CALITAS_BINNED"] = "0"
            else:
                os.environ.pop("CALITAS_BINNED", None)
            times = []
            for i in range(calls + 5):
                t0 = time.perf_counter()
                ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")
                if i >= 5:
                    times.append((time.perf_counter() - t0) * 1e3)
            times.sort()
            tm = ctx.timing()
            print("scale %-6g %-8s median %.3f ms  min %.3f ms  lanes %d binned %d  scan %.3f align+trace %.3f rows-kernel %.3f copy %.3f (sums, ms)  rows %d" % (
                scale, mode, times[len(times) // 2], times[0], tm["lanes"], tm["binned_lanes"], tm["scan_kernel_ms"], tm["align_kernel_ms"],
                tm["hits_kernel_ms"], tm["hits_copy_ms"], tm["hit_rows"]), flush=True)
        os.environ.pop("CALITAS_BINNED", None)
        ctx.unpin_host(buf.ctypes.data)
        ctx.close()


if __name__ == "__main__":
    main()
