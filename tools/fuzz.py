#!/usr/bin/env python3
"""Randomised parity: random genomes, guides (3' / 5' / no PAM, IUPAC codes, auxiliary PAMs), limits, costs and window sizes;
calitas_search_hits (one pass and lanes) against the CPU oracle, every column.  Usage: python3 tools/fuzz.py [iterations] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import calitas_amd as C
from calitas_amd import synth
import oracle_lib as O
from fasta_util import write_fasta

def run(iters, seed):
    rng = np.random.default_rng(seed)
    SKIP = {"aligner_version", "time_stamp"}
    IUPAC = "ACGTRYKMSWBDHVN"
    tmp = "/tmp/calitas_fuzz"
    os.makedirs(tmp, exist_ok=True)
    bad = 0
    t0 = time.time()
    for it in range(iters):
        L = int(rng.integers(14, 27))
        proto = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, L))
        if rng.random() < 0.15:
            k = int(rng.integers(0, L)); proto = proto[:k] + "RYSWN"[int(rng.integers(0, 5))] + proto[k + 1:]
        kind = int(rng.integers(0, 3))
        plen = int(rng.integers(2, 6))
        pam = "".join(IUPAC[int(x)] for x in rng.integers(0, len(IUPAC), plen)).lower() if kind else ""
        guide = proto + pam if kind != 2 else pam + proto
        aux = []
        if kind and rng.random() < 0.3:
            aux = ["".join(IUPAC[int(x)] for x in rng.integers(0, len(IUPAC), int(rng.integers(2, 6)))).lower()]
        d, p, g = int(rng.integers(0, 8 if L <= 18 else 6)), int(rng.integers(0, 3)), int(rng.integers(0, 5))
        Ov = int(rng.integers(1, 30))
        W = int(rng.choice([150, 400, 1000]))
        D = None if rng.random() < 0.6 else int(rng.integers(max(0, d - 1), d + g + p + 1))
        costs = {}
        if rng.random() < 0.25:
            costs = dict(guide_mismatch_net_cost=-int(rng.integers(60, 140)), pam_mismatch_net_cost=-int(rng.integers(100, 300)),
                         genome_gap_net_cost=-int(rng.integers(60, 140)), guide_gap_net_cost=-int(rng.integers(60, 140)))
        if W - (len(guide) + d + g - 1) <= 0:
            continue
        n_ctg = int(rng.integers(1, 4))
        spec = [("c%d" % i, int(rng.integers(300, 30000))) for i in range(n_ctg)]
        G = C.Guide(guide, aux)
        names, seqs = synth.make_genome(spec, int(rng.integers(0, 1 << 30)), guides=[(G.guide, G.pams[0] if G.pams else "", G.pam_is_five_prime)],
                                        sites_per_guide=int(rng.integers(5, 60)), softmask=float(rng.random() * 0.5), tandem_frac=float(rng.random() * (0.4 if rng.random() < 0.2 else 0.05)),
                                        n_run_ends=int(rng.integers(0, 200)), n_block=int(rng.integers(0, 1500)), step_hint=W - (len(guide) + d + g - 1))
        fa = write_fasta(os.path.join(tmp, "f.fa"), [(n, s.tobytes().decode()) for n, s in zip(names, seqs)])
        sw = int(rng.choice([0, 0, 0, 1, 2, 3]))         # the two unpinned readings of fgbio (oracle bits: 1 = per matrix, 2 = '='/'X' by score)
        okw = dict(window_size=W, d=d, p=p, g=g, D=-1 if D is None else D, O=Ov, threads=8, switches=sw)
        for k, o in (("guide_mismatch_net_cost", "m"), ("pam_mismatch_net_cost", "M"), ("genome_gap_net_cost", "b"), ("guide_gap_net_cost", "B")):
            if k in costs: okw[o] = costs[k]
        if os.environ.get("CALITAS_FUZZ_ONLY") and it != int(os.environ["CALITAS_FUZZ_ONLY"]):
            continue                                      # reproduce one iteration of a seed
        if os.environ.get("CALITAS_FUZZ_SW"):
            sw = int(os.environ["CALITAS_FUZZ_SW"]); okw["switches"] = sw
        if os.environ.get("CALITAS_FUZZ_LOG"):
            with open(os.environ["CALITAS_FUZZ_LOG"], "a") as lf:
                lf.write("iter %d guide %s aux %s d%d p%d g%d O%d W%d D%s sw%d costs %s spec %s\n" % (it, guide, aux, d, p, g, Ov, W, D, sw, costs, spec))
        try:
            _, want, _ = O.search_reference(fa, guide, "a", aux=aux, **okw)
        except RuntimeError as e:
            print("iter %d: oracle declined (%s)" % (it, e)); continue
        ctx = C.Context(0)
        ctx.set_reference_fasta(fa)
        try:
            params = C.make_params(window_size=W, max_guide_diffs=d, max_pam_mismatches=p, max_gaps_between_guide_and_pam=g, max_total_diffs=D,
                                   max_overlap=Ov, eqx_by_score=(1 if sw & 2 else 0) | (2 if sw & 1 else 0), **costs)
            res = {}
            for chunks in ("1", "2"):
                os.environ["CALITAS_CHUNKS"] = chunks
                try:
                    text, n = ctx.search_hits(G, "a", params, "v", "t")
                except C.CalitasError as e:
                    res[chunks] = "declined: %s" % e; continue
                res[chunks] = [{k: v for k, v in r.items() if k not in SKIP} for r in C.read_hits(text)]
        finally:
            ctx.close()
        w = [{k: v for k, v in r.items() if k not in SKIP} for r in want]
        ok = all(isinstance(v, str) or v == w for v in res.values())
        declined = [v for v in res.values() if isinstance(v, str)]
        if not ok:
            bad += 1
            print("MISMATCH iter %d guide %s aux %s d%d p%d g%d O%d W%d D%s sw%d costs %s: oracle %d rows, product %s" % (
                it, guide, aux, d, p, g, Ov, W, D, sw, costs, len(w), {k: (len(v) if not isinstance(v, str) else v) for k, v in res.items()}), flush=True)
        elif declined and it < 400:
            print("iter %d guide %s d%d: %s" % (it, guide, d, declined[0][:100]), flush=True)
    print("fuzz: %d iterations, %d mismatches, %.1f s" % (iters, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
