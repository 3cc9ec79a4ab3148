"""BASELINE config 5 (SearchReference --variants: PAM-less 20 nt guide, max-guide-diffs 8, a VCF at one variant per kilobase) on the
bench's own genome: 16 Mb with its VCF row by row against the oracle, a twentieth of the genome on both merge paths byte for byte, and
the call at its stated size (3.09 Gb, 3.0e6 variants, 4.1e7 rows / 21.8 GB of text) through the properties the reference's
post-processing guarantees (oracle/check_hits.cpp: ReferenceHit.sort order RH:284, removeOverlaps SR:653-675 per
chromosome : strand : variant_description group, 34 columns) and the row count of the bench line."""
import ctypes
import json
import os
import sys
import zlib

import pytest

import oracle_lib as O
from fasta_util import write_fasta

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = {"aligner_version", "time_stamp"}
KW = dict(max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
ROWS_AT_FULL_SIZE = 41_123_141          # hits_per_pass of `bench.py --config 5 --scale 1.0` (profiles/r03_bench_config5_full.json, r04_*)


@pytest.fixture(scope="module")
def bench_mod():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _search_variants(C, ctx, guide, vcf, keep=False):
    """calitas_search_variants (vcf_id NULL: the library computes the VCF's name:md5 itself); keep: the text stays in the library's block -> (address, bytes, rows, windows), to be freed by the caller."""
    lib = C._lib.lib
    g = C.Guide(guide).to_c()
    params = C.make_params(**KW)
    tsv, nbytes, rows, nwin = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    C._lib.check(ctx._h, lib.calitas_search_variants(ctx._h, ctypes.byref(g), b"c5", ctypes.byref(params), vcf.encode(), None, None, b"v0", b"stamp",
                                                     ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(rows), ctypes.byref(nwin)))
    if keep:
        return tsv, nbytes.value, rows.value, nwin.value
    try:
        return ctypes.string_at(tsv.value, nbytes.value), rows.value, nwin.value
    finally:
        lib.calitas_free(tsv)


def test_config5_16mb_with_its_vcf_against_the_oracle(bench_mod, tmp_path):
    import torch
    import calitas_amd as C
    names, seqs = bench_mod.build_genome(1.0, torch.device("cuda", 0), contig_indices=[0], guides=[bench_mod.GUIDE0], log=None)
    take = 16_000_000
    seq = bytes(seqs[0][:take])
    del seqs
    fa = write_fasta(str(tmp_path / "c5.fa"), [(names[0], seq.decode())])
    vcf = str(tmp_path / "c5.vcf")
    n_var = bench_mod.synthetic_vcf(vcf, [names[0]], [seq])
    guide = bench_mod.GUIDE0[:20]
    _, want, _ = O.search_reference_vcf(fa, vcf, guide, "c5", d=8, p=0, g=3, threads=16)
    ctx = C.Context(0)
    try:
        ctx.set_reference([names[0]], [seq], genome_build="testassembly")
        text, rows, nwin = _search_variants(C, ctx, guide, vcf)
        assert ctx.timing()["contig_passes"] == 1                 # the reference rows were built on the device, the variant windows' hits among them
    finally:
        ctx.close()
    got = C.read_hits(text.decode())

    def norm(rs):
        out = []
        for r in rs:
            r = {k: v for k, v in r.items() if k not in SKIP}
            if r.get("variant_vcf"):
                r["variant_vcf"] = r["variant_vcf"].split(":")[0]
            out.append(json.dumps(r, sort_keys=True))
        return sorted(out)
    assert n_var > 10000 and nwin >= n_var and rows == len(got) == len(want) > 150000
    assert sum(1 for r in got if r["variant_description"]) > 2000
    assert norm(got) == norm(want)      # as a multiset: ties between a variant group and the reference group have no pinned order (SR:656)
    assert O.check_hits_text(text, names, 10, 16) == dict(rows=rows, rows_with_variant=sum(1 for r in got if r["variant_description"]),
                                                          out_of_order=0, overlapping=0, malformed=0, threads=16)


def test_config5_twentieth_both_merge_paths_same_bytes(bench_mod, tmp_path, monkeypatch):
    """154 Mb, 25 contigs: the reference rows on the device with the variant windows' hits as key-only entries of its walk (the default)
    against the merge of alignment records on the host (CALITAS_VARIANTS_HOST=1) -- the same bytes."""
    import torch
    import calitas_amd as C
    names, seqs = bench_mod.build_genome(0.05, torch.device("cuda", 0), contig_indices=None, guides=[bench_mod.GUIDE0], log=None)
    vcf = str(tmp_path / "c5s.vcf")
    bench_mod.synthetic_vcf(vcf, names, seqs)
    ctx = C.Context(0)
    try:
        ctx.set_reference(names, seqs, genome_build="synthetic-hg38-sized")
        dev, rows, nwin = _search_variants(C, ctx, bench_mod.GUIDE0[:20], vcf)
        assert ctx.timing()["contig_passes"] == 25
        # the same call with its tables handed back before it returns (default: on the library's own thread, calitas_reap_wait waits)
        monkeypatch.setenv("CALITAS_FREE_NOW", "1")
        again, rows_a, nwin_a = _search_variants(C, ctx, bench_mod.GUIDE0[:20], vcf)
        monkeypatch.delenv("CALITAS_FREE_NOW")
        C._lib.lib.calitas_reap_wait()
        assert (zlib.crc32(again), len(again), rows_a, nwin_a) == (zlib.crc32(dev), len(dev), rows, nwin)
        monkeypatch.setenv("CALITAS_VARIANTS_HOST", "1")
        host, rows_h, nwin_h = _search_variants(C, ctx, bench_mod.GUIDE0[:20], vcf)
        assert ctx.timing()["contig_passes"] == 0
    finally:
        ctx.close()
    assert (zlib.crc32(dev), len(dev), rows, nwin) == (zlib.crc32(host), len(host), rows_h, nwin_h) and rows > 1_500_000
    assert O.check_hits_text(dev, names, 10, 16)["out_of_order"] == 0


def test_config5_at_its_stated_size_through_properties(bench_mod, tmp_path):
    import torch
    import calitas_amd as C
    names, seqs = bench_mod.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench_mod.GUIDE0], log=None)
    vcf = "/dev/shm/calitas_test_c5_%d.vcf" % os.getpid()
    ctx = C.Context(0)
    try:
        n_var = bench_mod.synthetic_vcf(vcf, names, seqs)
        ctx.set_reference(names, seqs, genome_build="synthetic-hg38-sized")
        del seqs
        tsv, nbytes, rows, nwin = _search_variants(C, ctx, bench_mod.GUIDE0[:20], vcf, keep=True)
        try:
            tm = ctx.timing()
            assert tm["contig_passes"] == 25                      # one device pass per contig, none declined to the host merge
            assert n_var > 3_000_000 and nwin >= n_var
            assert rows == ROWS_AT_FULL_SIZE and nbytes > 21_000_000_000
            got = O.check_hits_text(tsv.value, names, 10, 16, nbytes=nbytes)
            assert got["rows"] == rows and got["out_of_order"] == 0 and got["overlapping"] == 0 and got["malformed"] == 0
            assert got["rows_with_variant"] > 500_000
            # the same search into a page-locked block of the caller's (calitas_search_variants_into: every contig's rows straight to their
            # place over the bus, the kept variant-window rows written into their holes by the filler stage behind the VCF's MD5): same bytes
            import ctypes
            import numpy as np
            cap = nbytes + (1 << 20)
            addr = C.Context.alloc_host(cap)
            try:
                nb2, rows2, nwin2 = ctx.search_variants_into(C.Guide(bench_mod.GUIDE0[:20]), "c5", C.make_params(**KW), vcf, addr, cap, "v0", "stamp")
                assert (nb2, rows2, nwin2) == (nbytes, rows, nwin)
                a = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(tsv.value))
                b = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(addr))
                step = 1 << 30
                assert all(np.array_equal(a[o:o + step], b[o:o + step]) for o in range(0, nbytes, step))
                del a, b
            finally:
                C.Context.free_host(addr)
        finally:
            C._lib.lib.calitas_free(tsv)                          # (a block of 22 GB: parked for the next such search ...)
            C._lib.lib.calitas_release_parked()                   # (... which is not coming: back to the system)
    finally:
        ctx.close()
        if os.path.exists(vcf):
            os.remove(vcf)
