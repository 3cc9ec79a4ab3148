"""The last range's text is written to its final place in host memory by the rows kernel (no copy): many calls, every text's CRC against
the first call's, on slices and on window ranges -- a visibility problem (rows not yet in host memory when the call returns) would show
as a changing CRC.  python tools/inplace_stress.py [calls]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import calitas_amd as C

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
lengths = [len(s) for s in seqs]
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="synthetic")
del seqs
from calitas_amd import shard
G = C.Guide(bench.GUIDE0)
buf = np.zeros(128 << 20, dtype=np.uint8)
ctx.pin_host(buf.ctypes.data, buf.nbytes)
base = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
bad = 0
for n in (8, 4, 2):
    parts = shard.window_partition(lengths, n, 971)
    for r in (0, n - 1):
        p = C.make_params(first_window=parts[r][0], n_windows=parts[r][1], **base)
        os.environ["CALITAS_TEXT_IN_PLACE_OFF"] = "1"
        nb, rows = ctx.search_hits_into(G, "a", p, buf.ctypes.data, buf.nbytes, "v0", "stamp")
        want = zlib.crc32(bytes(buf[:nb]))
        del os.environ["CALITAS_TEXT_IN_PLACE_OFF"]
        t0 = time.time()
        for i in range(calls):
            buf[:nb] = 0                                                    # a stale buffer cannot pass
            nb2, rows2 = ctx.search_hits_into(G, "a", p, buf.ctypes.data, buf.nbytes, "v0", "stamp")
            got = zlib.crc32(bytes(buf[:nb2]))
            if (nb2, rows2, got) != (nb, rows, want):
                bad += 1
                print("MISMATCH rank %d of %d call %d: %d bytes %d rows crc %08x, want %d %d %08x" % (r, n, i, nb2, rows2, got, nb, rows, want), flush=True)
        print("rank %d of %d: %d calls, %d rows, %d bytes, crc %08x, lanes %d, %.1f s" % (r, n, calls, rows, nb, want, ctx.timing()["lanes"], time.time() - t0), flush=True)
print("inplace_stress: %d mismatches" % bad)
ctx.unpin_host(buf.ctypes.data)
ctx.close()
sys.exit(1 if bad else 0)
