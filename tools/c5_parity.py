#!/usr/bin/env python3
"""BASELINE config-5 shape at test size: PAM-less 20-mer, max-guide-diffs 8, with a VCF -- calitas_search_variants against the oracle
(multiset of rows: ties between a variant group and the reference group are unordered in the reference, SearchReference.scala:656).
Usage: python3 tools/c5_parity.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import calitas_amd as C
from calitas_amd import synth
import oracle_lib as O
from fasta_util import write_fasta
from test_oracle_variants import write_vcf
guide = "GTGACTTGAAGTCTCAGTAT"
rng = np.random.default_rng(5)
names, seqs = synth.make_genome([("chr1", 30000), ("chr2", 9000)], seed=3, guides=[(guide, "", False)], sites_per_guide=20, n_run_ends=20, n_block=200, softmask=0.2)
tmp = "/tmp/c5p"; os.makedirs(tmp, exist_ok=True)
fa = write_fasta(tmp + "/g.fa", [(n, s.tobytes().decode()) for n, s in zip(names, seqs)])
variants, afs = [], []
for name, s in zip(names, seqs):
    U = s.tobytes().decode().upper(); pos = 100
    while pos < len(U) - 100:
        pos += int(rng.integers(200, 1200))
        if pos >= len(U) - 10 or U[pos-1] not in "ACGT": continue
        rb = U[pos-1]; others = [b for b in "ACGT" if b != rb]
        variants.append((name, pos, "rs%d" % len(variants), rb, [others[0]])); afs.append([0.1])
vcf = write_vcf(tmp + "/v.vcf", variants, afs)
sr = C.SearchReference(guide=guide, guide_id="c5", ref=fa, variants=vcf, max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
text, n = sr.run("v", "t")
got = C.read_hits(text)
_, want, _ = O.search_reference_vcf(fa, vcf, guide, "c5", d=8, p=0, g=3)
SK = {"aligner_version", "time_stamp"}
def norm(rows):
    out = []
    for r in rows:
        r = {k: v for k, v in r.items() if k not in SK}
        if r.get("variant_vcf"): r["variant_vcf"] = r["variant_vcf"].split(":")[0]
        out.append(r)
    return out
key = lambda r: json.dumps(r, sort_keys=True)
a, b = sorted(map(key, norm(got))), sorted(map(key, norm(want)))
print("variants", len(variants), "windows", sr.variant_windows, "rows", n, "oracle", len(want), "equal multiset:", a == b, "with variants:", sum(1 for r in got if r["variant_id"]))
sys.exit(0 if a == b else 1)
