"""Config-5-shaped call (PAM-less d = 8 + synthetic VCF) at a small scale: writes the text to OUT, so that two builds of the library
(CALITAS_LIB_PATH) can be diffed.  python tools/c5_diff.py SCALE OUT"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import calitas_amd as C
from calitas_amd import _lib
lib = _lib.lib
scale, out = float(sys.argv[1]), sys.argv[2]
names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
ctx = C.Context(0)
ctx.set_reference(names, seqs, genome_build="synthetic")
vcf = "/tmp/c5_diff_%d.vcf" % os.getpid()
n = bench.synthetic_vcf(vcf, names, seqs)
params = C.make_params(max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
g = C.Guide(bench.GUIDE0[:20]).to_c()
tsv, nbytes, rows, nwin = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
_lib.check(ctx._h, lib.calitas_search_variants(ctx._h, ctypes.byref(g), b"bench", ctypes.byref(params), vcf.encode(), None, b"x:0", b"v", b"t",
                                               ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(rows), ctypes.byref(nwin)))
text = ctypes.string_at(tsv.value, nbytes.value)
lib.calitas_free(tsv)
open(out, "wb").write(text)
print(n, "variants", rows.value, "rows", nwin.value, "windows", len(text), "bytes")
