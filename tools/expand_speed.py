"""Host half of the compact rows alone: calitas_expand_rows on a text shaped like an hg38-sized call's first range (118 000 rows of ~100
compact bytes, 172 bytes of head + tail), ms per call and GB/s of output by worker count: python tools/expand_speed.py [rows] [repeats]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import calitas_amd as C
from calitas_amd import _lib

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 118000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(5)
head = b"guide-0\tGAGTCCGAGCAGAAGAAGAA\tsynthetic\t"
tail = b"CALITAS:SearchReference\tv0\tNRG\t" + b"max-guide-diffs=5;max-pam-mismatches=1;max-gaps-between-guide-and-pam=2;max-overlap=10;window-size=1000" + b"\tstamp-2026-01-01\n"
rows = [b"chr%d\t%d\t%s" % (int(rng.integers(1, 23)), int(rng.integers(0, 10 ** 8)), b"x" * int(rng.integers(70, 110))) for _ in range(2000)]
compact = b"".join(rows[i % 2000] + b"\n" for i in range(n_rows))
out_len = len(compact) + n_rows * (len(head) + len(tail) - 1)
out = np.zeros(out_len + 64, dtype=np.uint8)
ctx = C.Context(-1)
written = ctypes.c_uint64()
print("compact %.1f MB -> %.1f MB, %d rows; pool of %s" % (len(compact) / 1e6, out_len / 1e6, n_rows, os.environ.get("CALITAS_THREADS", "default")))
for threads in (1, 2, 4, 8, 16):
    os.environ["CALITAS_EXPAND_THREADS"] = str(threads)
    times = []
    for i in range(reps + 3):
        t = time.perf_counter()
        rc = _lib.lib.calitas_expand_rows(ctx._h, compact, len(compact), n_rows, head, tail, out.ctypes.data, out_len, ctypes.byref(written))
        if i >= 3:
            times.append((time.perf_counter() - t) * 1e3)
        assert rc == 0 and written.value == out_len
    times.sort()
    print("threads %2d: median %.3f ms  min %.3f  max %.3f   %.1f GB/s out" % (threads, times[len(times) // 2], times[0], times[-1], out_len / times[len(times) // 2] / 1e6))
ctx.close()
