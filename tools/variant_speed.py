#!/usr/bin/env python3
"""How fast is the variant branch (SearchReference --variants)?  Synthetic genome of the given size with one biallelic SNV /
small indel per `spacing` bases.  Usage: python3 tools/variant_speed.py [mb] [spacing]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import calitas_amd as C
from calitas_amd import synth

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
spacing = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = int(mb * 1e6)
rng = np.random.default_rng(0xC5)
seq = synth.random_bases(rng, n, gc=0.41)
tmp = "/tmp/variant_speed"
os.makedirs(tmp, exist_ok=True)
with open(tmp + "/ref.fa", "w") as f:
    f.write(">chr1\n")
    s = seq.tobytes().decode()
    for i in range(0, n, 100):
        f.write(s[i:i + 100] + "\n")
with open(tmp + "/v.vcf", "w") as f:
    f.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"AF\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
    k = 0
    for pos in range(spacing, n - spacing, spacing):
        p = pos + int(rng.integers(0, spacing // 2))
        ref = s[p - 1]
        kind = int(rng.integers(0, 10))
        if kind < 8:
            alt = "ACGT"[("ACGT".index(ref) + 1 + int(rng.integers(0, 3))) % 4]
        elif kind == 8:
            alt = ref + "ACGT"[int(rng.integers(0, 4))] * int(rng.integers(1, 4))
        else:
            ref = s[p - 1:p + int(rng.integers(1, 4))]; alt = ref[0]
        f.write("chr1\t%d\trs%d\t%s\t%s\t.\tPASS\tAF=%.3f\n" % (p, k, ref, alt, float(rng.uniform(0.01, 0.5))))
        k += 1
print("genome %d bp, %d variants" % (n, k), flush=True)
ctx = C.Context(0)
ctx.set_reference_fasta(tmp + "/ref.fa")
for guide, kw in (("CTTGCCCCACAGGGCAGTAAnrg", dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)),):
    t = time.perf_counter()
    sr = C.SearchReference(guide=guide, guide_id="v", context=ctx, variants=tmp + "/v.vcf", **kw)
    text, rows = sr.run("v0", "stamp")
    dt = time.perf_counter() - t
    nv = sum(1 for ln in text.splitlines()[1:] if ln.split("\t")[11])
    print("%s: %.2f s, %d rows (%d with variants): %.0f variants/s" % (guide, dt, rows, nv, k / dt), flush=True)
ctx.close()
