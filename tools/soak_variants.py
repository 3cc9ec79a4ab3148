#!/usr/bin/env python3
"""Soak of the variant search at BASELINE config 5's size: the same call again and again -- into a block of the library's, into a
page-locked block of the caller's, with the kept rows sent to the device (CALITAS_VARIANTS_ROWS=device) -- every text's CRC must be
the first one's (the stages of the call are seven threads and a worker pool: a race shows as a different byte sooner or later).
Usage: python3 tools/soak_variants.py [calls] [scale]"""
import ctypes, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import calitas_amd as C

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
vcf = "/dev/shm/calitas_soak_%d.vcf" % os.getpid()
n_var = bench.synthetic_vcf(vcf, names, seqs)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="soak")
del seqs
params = C.make_params(max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
G = C.Guide(bench.GUIDE0[:20])


def crc_of(addr, n):
    a = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(addr))
    c = 0
    for o in range(0, n, 1 << 28):
        c = zlib.crc32(a[o:o + (1 << 28)], c)
    return c


lib = C._lib.lib
g = G.to_c()
tsv, nb, rows, nwin = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
C._lib.check(ctx._h, lib.calitas_search_variants(ctx._h, ctypes.byref(g), b"soak", ctypes.byref(params), vcf.encode(), None, None, b"v", b"t",
                                                 ctypes.byref(tsv), ctypes.byref(nb), ctypes.byref(rows), ctypes.byref(nwin)))
want = (nb.value, rows.value, crc_of(tsv.value, nb.value))
lib.calitas_free(tsv)
print("reference call: %d bytes, %d rows, %d variants, crc %08x" % (want[0], want[1], n_var, want[2]), flush=True)
cap = want[0] + (1 << 20)
addr = C.Context.alloc_host(cap)
bad = 0
t0 = time.time()
try:
    for k in range(calls):
        mode = ("into", "library", "device-rows")[k % 3]
        if mode == "device-rows":
            os.environ["CALITAS_VARIANTS_ROWS"] = "device"
        else:
            os.environ.pop("CALITAS_VARIANTS_ROWS", None)
        t = time.time()
        if mode == "library":
            C._lib.check(ctx._h, lib.calitas_search_variants(ctx._h, ctypes.byref(g), b"soak", ctypes.byref(params), vcf.encode(), None, None, b"v", b"t",
                                                             ctypes.byref(tsv), ctypes.byref(nb), ctypes.byref(rows), ctypes.byref(nwin)))
            dt = time.time() - t
            got = (nb.value, rows.value, crc_of(tsv.value, nb.value))
            lib.calitas_free(tsv)
        else:
            n, r, _ = ctx.search_variants_into(G, "soak", params, vcf, addr, cap, "v", "t")
            dt = time.time() - t
            got = (n, r, crc_of(addr, n))
        ok = got == want
        bad += 0 if ok else 1
        print("call %2d (%s): %.3f s  %s" % (k, mode, dt, "same text" if ok else "DIFFERENT: %r" % (got,)), flush=True)
finally:
    C.Context.free_host(addr)
    ctx.close()
    os.remove(vcf)
print("soak_variants: %d calls, %d with a different text, %.0f s" % (calls, bad, time.time() - t0))
sys.exit(1 if bad else 0)
