"""Stability of repeated calls: N calitas_search_hits_into calls on the bench genome, every text compared (CRC) with the first one;
prints the time of the first / median / last hundred calls and the process's resident set at both ends.
python tools/repeat_calls.py [scale] [calls]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0
    return 0.0


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import numpy as np, torch, bench
    import calitas_amd as C
    names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="synthetic")
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = C.Guide(bench.GUIDE0)
    buf = np.zeros(256 << 20, dtype=np.uint8)
    ctx.pin_host(buf.ctypes.data, buf.nbytes)
    first, times, bad = None, [], 0
    r0 = rss_mb()
    for i in range(calls):
        t0 = time.perf_counter()
        nbytes, rows = ctx.search_hits_into(G, "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")
        times.append((time.perf_counter() - t0) * 1e3)
        if i % 10 == 0 or i < 5:
            crc = (zlib.crc32(buf[:nbytes]), nbytes, rows)
            if first is None:
                first = crc
            bad += crc != first
    med = lambda v: sorted(v)[len(v) // 2]
    print("%d calls: first hundred %.3f ms, all %.3f ms, last hundred %.3f ms (medians); %d mismatching texts; RSS %.0f -> %.0f MB" % (
        calls, med(times[5:105]), med(times), med(times[-100:]), bad, r0, rss_mb()))
    ctx.unpin_host(buf.ctypes.data)
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
