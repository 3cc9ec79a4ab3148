"""CPU tests of the product's host side (no GPU): the C ABI loads and exports every declared symbol, the packed
reference round-trips, the window table equals windowIterator, and the two host stages above the kernels
(per-window filter, removeOverlaps/sort/rows) reproduce the oracle when fed the oracle's own alignments.
The oracle is used here only as the checker / input generator."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from fasta_util import write_fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP_COLS = {"aligner_version", "time_stamp"}


@pytest.fixture(scope="module")
def C():
    import calitas_amd
    return calitas_amd


def test_abi_exports_every_declared_symbol(C):
    header = open(os.path.join(ROOT, "include", "calitas_hip.h")).read()
    declared = set(re.findall(r"\b(calitas_[a-z_]+)\s*\(", header))
    declared -= {"calitas_ctx"}
    assert declared == set(C._lib.SYMBOLS), declared ^ set(C._lib.SYMBOLS)
    for s in declared:
        assert getattr(C._lib.lib, s) is not None
    assert b"gfx950" in C._lib.lib.calitas_version()


def test_no_device_fails_loudly(C):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(C.CalitasError) as e:
        C.Context(0)
    assert e.value.code == C._lib.ENODEV
    ctx = C.Context(-1)
    ctx.set_reference(["c"], [b"ACGT" * 100])
    with pytest.raises(C.CalitasError) as e:
        ctx.search([C.Guide("ACGTACGTACGTACGTACGTnrg")], C.make_params())
    assert e.value.code == C._lib.ENODEV and "no CPU fallback" in str(e.value)


def test_guide_parsing_matches_reference_rules(C):
    g = C.Guide("AACCAACCAACCnrg")
    assert (g.guide, g.pams, g.pam_is_five_prime) == ("AACCAACCAACC", ["nrg"], False)
    g = C.Guide("tttvAACCAACCAACC")
    assert (g.guide, g.pams, g.pam_is_five_prime) == ("AACCAACCAACC", ["tttv"], True)
    g = C.Guide("AACCGGTTACGTnnn", ["nnnn", "nn"])
    assert g.pams == ["nnn", "nnnn", "nn"] and g.pam_length == 4 and g.length == 16
    assert C.Guide("GTGACTTGAAGTCTCAGTATA").pams == []
    for bad, aux in (("acgt", ()), ("ACGTacgtACGT", ()), ("ACGT", ("nrg",)), ("ACGTnrg", ("NGG",))):
        with pytest.raises(ValueError):
            C.Guide(bad, aux)


def test_packed_reference_roundtrip_and_runs(C):
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"ACGTacgtNnRYKMSWBDHVUu*", dtype=np.uint8)
    probs = np.array([20, 20, 20, 20, 5, 5, 5, 5, 4, 1] + [0.3] * 12 + [0.2])
    probs = probs / probs.sum()
    seqs = []
    for n in (5000, 333, 1, 70000):
        s = alphabet[rng.choice(len(alphabet), size=n, p=probs)]
        if n > 1000:
            s[100:400] = ord("N")
            s[-50:] = ord("N")
        seqs.append(s)
    ctx = C.Context(-1)
    ctx.set_reference(["a", "b", "c", "d"], seqs)
    assert ctx.contig_lengths == [5000, 333, 1, 70000]
    for i, s in enumerate(seqs):
        want = s.tobytes().decode().upper()
        got = "".join(ctx.fetch_bases(i, st, min(997, len(s) - st)) for st in range(0, len(s), 997))
        assert got == want


@pytest.mark.parametrize("window,step", [(1000, 971), (451, 426), (120, 91), (50, 7)])
def test_window_table_equals_window_iterator(C, window, step, tmp_path):
    rng = np.random.default_rng(window)
    contigs = []
    for i, n in enumerate((4321, 999, 1000, 1001, 972, 30, 2, 1, 2600)):
        s = np.frombuffer(b"ACGTacgtn", dtype=np.uint8)[rng.integers(0, 9, size=n)].copy()
        if n > 500:
            s[:137] = ord("N")
            s[n // 2: n // 2 + 300] = ord("N")
            s[-61:] = ord("N")
        if i == 8:
            s[1000:2100] = ord("N")       # windows that are entirely N
        contigs.append(("c%d" % i, s.tobytes().decode()))
    fa = write_fasta(str(tmp_path / "w.fa"), contigs)
    ctx = C.Context(-1)
    ctx.set_reference_fasta(fa)
    min_len = 23
    want = [(n, s - 1, e) for (n, s, e, ln) in O.windows(fa, window, step) if ln >= min_len]
    got = [(ctx.contig_names[c], a, b) for (c, a, b) in ctx.window_table(window, step, min_len)]
    assert got == want


def _oracle_alignments(C, guide, contig_name, contig_index, seq, params_kw, window_size=1000, window_range=None):
    """Per-window SequentialGuideAligner.align results of the ORACLE, converted to product Alignment records (all windows of the
    contig, or windows [window_range[0], window_range[1]) of it)."""
    G = C.Guide(guide)
    step = window_size - (G.cli_length + params_kw["d"] + params_kw["g"] - 1)
    out = []
    n = len(seq)
    for k, start in enumerate(range(0, n - 1, step)):
        if window_range is not None and not (window_range[0] <= k < window_range[1]):
            continue
        end = min(n, start + window_size)
        a, b = start, end
        while a < b and seq[a] == "N":
            a += 1
        while a < b and seq[b - 1] == "N":
            b -= 1
        if b - a < G.cli_length:
            continue
        rows = O.align(guide, seq[a:b].upper(), params_kw["d"], params_kw["g"], params_kw["p"], params_kw["D"], O=params_kw["O"],
                       off=a, name=contig_name)
        for r in rows:
            al = C.Alignment.__new__(C.Alignment)
            al.guide_index, al.contig_index, al.window_start = 0, contig_index, a
            al.start_offset, al.end_offset = r["start"], r["end"]
            al.guide_start_offset, al.guide_end_offset = r["gstart"], r["gend"]
            al.score, al.strand = r["score"], r["strand"]
            al.pam_index = 0 if G.pams else -1
            ops = ""
            for q, m in zip(r["padded_guide"], r["padded_alignment"]):
                ops += "=" if m == "|" else "X" if m == "." else ("D" if q == "-" else "I")
            al.ops = ops
            out.append(al)
    return out


@pytest.mark.parametrize("guide", ["CTTGCCCCACAGGGCAGTAAnrg", "tttvCTTGCCCCACAGGGCAGTAA", "CTTGCCCCACAGGGCAGTAA"])
def test_hits_tsv_stage_matches_oracle(C, guide, tmp_path):
    """removeOverlaps + sort + the 34-column rows (product host code) on the oracle's per-window alignments must give
    the oracle's hits.txt."""
    from calitas_amd import synth
    G = C.Guide(guide)
    pam = G.pams[0] if G.pams else ""
    names, seqs = synth.make_genome([("chrA", 30000), ("chrB", 12000)], seed=5, guides=[(G.guide, pam, G.pam_is_five_prime)],
                                    sites_per_guide=40, n_run_ends=150, n_block=1200, tandem_frac=0.05)
    contigs = [(n, s.tobytes().decode()) for n, s in zip(names, seqs)]
    fa = write_fasta(str(tmp_path / "h.fa"), contigs)
    kw = dict(d=4, p=1, g=2, D=7, O=10)
    alns = []
    for ci, (n, s) in enumerate(contigs):
        alns += _oracle_alignments(C, guide, n, ci, s, kw)
    ctx = C.Context(-1)
    ctx.set_reference_fasta(fa)
    params = C.make_params(max_guide_diffs=4, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2, max_total_diffs=7)
    text, n_rows = ctx.hits_tsv(G, "a", params, alns)
    got = C.read_hits(text)
    _, want, _ = O.search_reference(fa, guide, "a", d=4, p=1, g=2, D=7)
    strip = lambda rows: [{k: v for k, v in r.items() if k not in SKIP_COLS} for r in rows]
    assert len(want) > 5
    assert strip(got) == strip(want)
    # padded strings through the dedicated entry point
    pg, pa, pt = ctx.padded_strings(G, alns[0])
    assert len(pg) == len(pa) == len(pt) == len(alns[0].ops)


def test_window_filter_stage_matches_oracle(C):
    """The per-window greedy filter (SGA:315-320) of the product, fed every extended alignment of a window (the oracle
    run with limits that filter nothing), must keep what the oracle keeps with the real limits."""
    from calitas_amd import synth
    rng = np.random.default_rng(17)
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    unit = "CTTGCCCCACAGGGCAGTAATGG"
    # a window full of overlapping near-matches: tandem copies with mutations
    seq = "".join(synth.mutate(rng, unit, int(rng.integers(0, 4))) + "ACG"[: int(rng.integers(0, 3))] for _ in range(30))
    seq = (seq + synth.revcomp(seq))[:1000]
    for D, Ov in ((8, 10), (4, 10), (8, 0), (8, 100), (3, 5)):
        everything = O.align(guide, seq, 5, 3, 1, 10 ** 6, O=10 ** 6, off=500, name="w")
        want = O.align(guide, seq, 5, 3, 1, D, O=Ov, off=500, name="w")
        recs = []
        for r in everything:
            al = C.Alignment.__new__(C.Alignment)
            al.guide_index, al.contig_index, al.window_start = 0, 0, 500
            al.start_offset, al.end_offset, al.guide_start_offset, al.guide_end_offset = r["start"], r["end"], r["gstart"], r["gend"]
            al.score, al.strand, al.pam_index = r["score"], r["strand"], 0
            al.ops = "".join("=" if m == "|" else "X" if m == "." else ("D" if q == "-" else "I")
                             for q, m in zip(r["padded_guide"], r["padded_alignment"]))
            recs.append(al)
        kept = C.window_filter(recs, D, Ov)
        assert len(everything) > 20
        assert [(k.strand, k.start_offset, k.end_offset, k.score, k.ops) for k in kept] == \
               [(w["strand"], w["start"], w["end"], w["score"],
                 "".join("=" if m == "|" else "X" if m == "." else ("D" if q == "-" else "I")
                         for q, m in zip(w["padded_guide"], w["padded_alignment"]))) for w in want]


def test_cli_binary_exists_and_reports_usage():
    import subprocess
    exe = os.path.join(ROOT, "calitas_amd", "calitas")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "SearchReference" in r.stderr


def test_persistent_index_roundtrip(C, tmp_path):
    """calitas_save_index / calitas_load_index reproduce the packed reference bit for bit (window table, bases)."""
    from calitas_amd import synth
    names, seqs = synth.make_genome([("chrA", 70000), ("chrB", 5000), ("tiny", 9)], seed=3, n_run_ends=500, n_block=3000)
    seqs[1][100:110] = np.frombuffer(b"RYKMSWBDHV", dtype=np.uint8)
    a = C.Context(-1)
    a.set_reference(names, seqs, genome_build="hgTest")
    idx = str(tmp_path / "ref.calidx")
    a.save_index(idx)
    b = C.Context(-1)
    b.load_index(idx)
    assert (b.contig_names, b.contig_lengths, b.reference_info()) == (a.contig_names, a.contig_lengths, a.reference_info())
    for ci, s in enumerate(seqs):
        assert b.fetch_bases(ci, 0, len(s)) == s.tobytes().decode().upper()
    assert b.window_table(1000, 971, 23) == a.window_table(1000, 971, 23)
    with open(idx, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(C.CalitasError):
        C.Context(-1).load_index(idx)


def test_module_cli_parses_the_reference_flags():
    """python -m calitas_amd <Tool>: flag names as in the reference's tools (no GPU needed to parse)."""
    from calitas_amd.__main__ import main
    for tool in ("SearchReference", "AlignToReference", "PairwiseAlignSequences", "PrepareVcf"):
        with pytest.raises(SystemExit) as e:
            main([tool, "--help"])
        assert e.value.code == 0
    with pytest.raises(SystemExit) as e:
        main(["SearchReference", "-i", "ACGTnrg"])    # -I and -r are required (SearchReference.scala:452-455)
    assert e.value.code != 0


def test_scan_census_follows_the_kernel(tmp_path):
    """bench.py's roofline.valu.mix_limit prices the scan kernel's own instruction mix from profiles/r04_scan_census.json; the file is
    what tools/scan_census.py makes of the compiler's output for calitas_amd/csrc/scan_rows.hip today (a change to the kernel without a
    fresh census fails here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "census.json")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "scan_census.py"), "-o", out], stdout=subprocess.DEVNULL)
    fresh, kept = json.load(open(out)), json.load(open(os.path.join(root, "profiles", "r04_scan_census.json")))
    for k in ("kernel", "words_per_lane", "row_iteration_ns", "guide_strand_rest_ns", "once_per_wave_ns", "valu_per_row_iteration"):
        assert fresh[k] == kept[k], k
    row = kept["strands"][0]["row_iteration"]
    assert row["v_addc_co_u32"] + row.get("v_add_co_u32", 0) == kept["words_per_lane"]      # the carry chain: one add per word
    assert 9 <= kept["valu_per_row_iteration"][0] / kept["words_per_lane"] <= 12


def test_rank_cpus_from_a_sysfs_tree(tmp_path):
    """calitas_amd.shard.rank_cpus: a rank's CPUs = those of its GPU's NUMA node, shared among the ranks on that node (bench.py pins
    every rank of an N > 1 job before it starts a thread).  A fake /sys: 8 GPUs, four per socket, 32 CPUs per socket, two CPU-only
    KFD nodes in front (as on the MI355X boxes)."""
    from calitas_amd import shard
    root = tmp_path
    def put(rel, text):
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    for i in range(2):
        put("sys/class/kfd/kfd/topology/nodes/%d/properties" % i, "cpu_cores_count 32\nsimd_count 0\ndrm_render_minor -1\n")
    for g in range(8):
        put("sys/class/kfd/kfd/topology/nodes/%d/properties" % (2 + g), "cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % (128 + g))
        put("sys/class/drm/renderD%d/device/numa_node" % (128 + g), "%d\n" % (g // 4))
    put("sys/devices/system/node/node0/cpulist", "0-15,64-79\n")
    put("sys/devices/system/node/node1/cpulist", "16-31,80-95\n")
    assert shard.gpu_numa_nodes(str(root)) == [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = set(range(128))
    sets = [shard.rank_cpus(r, 8, allowed, str(root)) for r in range(8)]
    assert all(len(s) == 8 for s in sets) and len(set(c for s in sets for c in s)) == 64          # disjoint shares
    assert set(sets[0] + sets[1] + sets[2] + sets[3]) == set(range(0, 16)) | set(range(64, 80))
    assert set(sets[5]) <= set(range(16, 32)) | set(range(80, 96))
    # two ranks on one box: GPUs 0 and 1 share node 0 -> half of the node each; a cpuset that leaves a rank one CPU: left alone
    two = [shard.rank_cpus(r, 2, allowed, str(root)) for r in range(2)]
    assert len(two[0]) == len(two[1]) == 16 and not set(two[0]) & set(two[1])
    assert shard.rank_cpus(0, 8, {0, 1, 2, 3}, str(root)) is None
    # HIP_VISIBLE_DEVICES=4,5: rank 0 sits on GPU 4 = node 1
    assert set(shard.rank_cpus(0, 2, allowed, str(root), visible=[4, 5])) <= set(range(16, 32)) | set(range(80, 96))
    # no KFD topology (this container): nothing to do
    assert shard.gpu_numa_nodes(str(tmp_path / "nowhere")) == [] and shard.rank_cpus(0, 2, allowed, str(tmp_path / "nowhere")) is None


def test_expand_rows_puts_head_and_tail_back():
    """calitas_expand_rows (the host half of the compact rows a guide batch moves over PCIe): `chromosome \\t middle \\n` per row becomes
    head | chromosome \\t middle | tail, on one thread (small texts) and on the worker pool (rows cut at arbitrary byte boundaries between
    the workers); a text that does not hold the stated number of rows is refused."""
    import ctypes
    import numpy as np
    import calitas_amd as C
    from calitas_amd import _lib
    ctx = C.Context(-1)
    try:
        rng = np.random.default_rng(3)
        head, tail = b"guide-7\tACGTACGTACGTACGTACGT\tbuild-x\t", b"CALITAS:SearchReference\tv\tnrg\tmax-guide-diffs=5;window-size=1000\tstamp\n"
        for n_rows in (0, 1, 7, 40000):
            rows = [b"chr%d\t%d\t%s" % (int(rng.integers(1, 23)), int(rng.integers(0, 10 ** 8)), b"x" * int(rng.integers(1, 240))) for _ in range(n_rows)]
            compact = b"".join(r + b"\n" for r in rows)
            want = b"".join(head + r + tail for r in rows)
            out = ctypes.create_string_buffer(len(want) + 1)
            written = ctypes.c_uint64()
            rc = _lib.lib.calitas_expand_rows(ctx._h, compact, len(compact), n_rows, head, tail, out, len(want), ctypes.byref(written))
            assert rc == 0 and written.value == len(want) and out.raw[:len(want)] == want, n_rows
            if n_rows:
                assert _lib.lib.calitas_expand_rows(ctx._h, compact, len(compact), n_rows + 1, head, tail, out, len(want) + 200, ctypes.byref(written)) != 0
                assert _lib.lib.calitas_expand_rows(ctx._h, compact[:-1], len(compact) - 1, n_rows, head, tail, out, len(want), ctypes.byref(written)) != 0
                assert _lib.lib.calitas_expand_rows(ctx._h, compact, len(compact), n_rows, head, tail, out, len(want) - 1, ctypes.byref(written)) != 0
        # pieces of the expansion job (64 KB of compact text each, post.cpp RowExpansion): rows longer than a piece, rows that end on a
        # piece boundary, and every number of workers
        for trial, (n_rows, longest) in enumerate(((30000, 90), (2000, 3000), (300, 70000), (65536 // 8, 8), (9000, 300))):
            lens = rng.integers(1, longest, size=n_rows) if longest != 8 else np.full(n_rows, 5)   # (5 x's + "c\t" + newline = 8 bytes)
            rows = [b"c\t" + b"x" * int(l) for l in lens]
            compact = b"".join(r + b"\n" for r in rows)
            want = b"".join(head + r + tail for r in rows)
            out = ctypes.create_string_buffer(len(want) + 1)
            for threads in ("1", "2", "3", "16"):
                os.environ["CALITAS_EXPAND_THREADS"] = threads
                try:
                    ctypes.memset(out, 0, len(want) + 1)
                    rc = _lib.lib.calitas_expand_rows(ctx._h, compact, len(compact), n_rows, head, tail, out, len(want), ctypes.byref(written))
                    assert rc == 0 and written.value == len(want) and out.raw[:len(want)] == want, (trial, threads)
                    assert _lib.lib.calitas_expand_rows(ctx._h, compact, len(compact), n_rows - 1, head, tail, out, len(want), ctypes.byref(written)) != 0
                finally:
                    del os.environ["CALITAS_EXPAND_THREADS"]
    finally:
        ctx.close()


def test_callers_on_several_threads_share_the_worker_pool(C, tmp_path):
    """WorkerPool::run hands numbered shares to whichever threads are free and offer() hands out a job's pieces: callers on different
    threads run side by side on one context's pool (the variant branch does, four threads at a time).  Several threads at once: the
    rows stage (calitas_hits_tsv: pool sections) and the row expansion (an offered job), every result as from one thread alone."""
    import ctypes
    import threading
    from calitas_amd import synth, _lib
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    G = C.Guide(guide)
    names, seqs = synth.make_genome([("chrA", 40000), ("chrB", 15000)], seed=8, guides=[(G.guide, G.pams[0], G.pam_is_five_prime)],
                                    sites_per_guide=60, n_run_ends=100, n_block=1200, tandem_frac=0.05)
    contigs = [(n, s.tobytes().decode()) for n, s in zip(names, seqs)]
    fa = write_fasta(str(tmp_path / "p.fa"), contigs)
    kw = dict(d=4, p=1, g=2, D=7, O=10)
    alns = []
    for ci, (n, sq) in enumerate(contigs):
        alns += _oracle_alignments(C, guide, n, ci, sq, kw)
    ctx = C.Context(-1)
    ctx.set_reference_fasta(fa)
    params = C.make_params(max_guide_diffs=4, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2, max_total_diffs=7)
    rng = np.random.default_rng(12)
    head, tail = b"g\tACGT\tb\t", b"CALITAS\tv\tnrg\tparams\tstamp\n"
    rows = [b"c\t" + b"x" * int(l) for l in rng.integers(1, 200, size=30000)]
    compact = b"".join(r + b"\n" for r in rows)
    want_rows = b"".join(head + r + tail for r in rows)
    want_text, want_n = ctx.hits_tsv(G, "a", params, alns)
    os.environ["CALITAS_EXPAND_THREADS"] = "16"
    errors = []

    def tsv_worker():
        try:
            for _ in range(6):
                text, n = ctx.hits_tsv(G, "a", params, alns)
                assert n == want_n and text == want_text
        except Exception as e:       # noqa: BLE001  (reported by the main thread)
            errors.append(repr(e))

    def expand_worker():
        try:
            out = ctypes.create_string_buffer(len(want_rows) + 1)
            written = ctypes.c_uint64()
            for _ in range(12):
                rc = _lib.lib.calitas_expand_rows(ctx._h, compact, len(compact), len(rows), head, tail, out, len(want_rows), ctypes.byref(written))
                assert rc == 0 and written.value == len(want_rows) and out.raw[:len(want_rows)] == want_rows
        except Exception as e:       # noqa: BLE001
            errors.append(repr(e))

    try:
        threads = [threading.Thread(target=tsv_worker) for _ in range(2)] + [threading.Thread(target=expand_worker) for _ in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        del os.environ["CALITAS_EXPAND_THREADS"]
        ctx.close()
    assert not errors, errors


def test_every_environment_switch_is_in_the_table():
    """calitas_switches() (calitas_amd/csrc/tuning.hpp) lists every CALITAS_* switch the library reads, and nothing else: the sources
    read them through TUNE_GET / TUNE_ON / TUNE_SET only (a name missing from the table does not compile -- nothing in the library
    stops the process over a switch), and no std::getenv of a CALITAS_ name is left.  Switches that trade correct output for a timing
    experiment (CALITAS_BINNED_SKIP) exist in `make EXPERIMENTS=1` builds only: the shipped library does not list or read them."""
    import glob
    import re
    import calitas_amd  # noqa: F401
    from calitas_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    listed = {ln.split(" ", 1)[0] for ln in _lib.lib.calitas_switches().decode().splitlines() if ln}
    used, raw, aborts = set(), [], []
    for f in glob.glob(os.path.join(root, "calitas_amd", "csrc", "*.[ch]*")):
        if f.endswith(".o"):
            continue
        text = open(f, errors="replace").read()
        aborts += [os.path.basename(f)] if re.search(r"\babort\s*\(", re.sub(r"//[^\n]*", "", text)) and not f.endswith("dbg_alloc.hpp") else []
        if f.endswith("tuning.hpp"):
            continue
        text = re.sub(r"#ifdef CALITAS_EXPERIMENTS.*?#endif", "", text, flags=re.S)   # (not in the shipped build)
        used |= set(re.findall(r'TUNE_(?:GET|ON|SET)\("(CALITAS_[A-Z0-9_]+)"\)', text))
        raw += re.findall(r'[^:]getenv\("(CALITAS_[A-Z0-9_]+)"\)', text)
        assert not re.findall(r"tune::(?:get|on|is_set)\(", text), f
    assert not raw, raw
    assert not aborts, aborts                                  # the library reports, it does not stop the process
    assert used == listed, (sorted(used - listed), sorted(listed - used))
    assert "CALITAS_BINNED_SKIP" not in listed


def test_worker_threads_and_resident_contigs_of_a_rank(tmp_path):
    """What bench.py --gpus N gives every rank (calitas_amd/shard.py): worker threads from the rank's CPU list and its share of the
    cgroup's CPU quota (a GPU box bounds a process by cpu.max on a shared host, not by a CPU set), and -- window partition -- the contigs
    its range touches as the only ones whose bases it holds."""
    from calitas_amd import shard, synth
    assert shard.worker_threads(list(range(32)), 8, None) == 16 and shard.worker_threads(list(range(32)), 8, 16.0) == 2
    assert shard.worker_threads(list(range(6)), 2, 16.0) == 6 and shard.worker_threads(None, 4, 16.0) == 4 and shard.worker_threads([0], 1, 0.5) == 2
    (tmp_path / "proc/self").mkdir(parents=True)
    (tmp_path / "proc/self/cgroup").write_text("0::/a/b\n")
    (tmp_path / "sys/fs/cgroup/a/b").mkdir(parents=True)
    (tmp_path / "sys/fs/cgroup/a/b/cpu.max").write_text("max 100000\n")
    assert shard.cgroup_cpu_quota(str(tmp_path)) is None
    (tmp_path / "sys/fs/cgroup/a/cpu.max").write_text("1600000 100000\n")
    (tmp_path / "sys/fs/cgroup/cpu.max").write_text("3200000 100000\n")
    assert shard.cgroup_cpu_quota(str(tmp_path)) == 16.0                      # the tightest on the way up
    assert shard.cgroup_cpu_quota(str(tmp_path / "nowhere")) is None
    # hg38 on 8 ranks: every rank holds 2-7 contigs (13-20 % of the bases, against an eighth owned), all 25 are held by somebody, and a rank's share is a fraction of the genome
    L = synth.HG38_LENGTHS
    held = [shard.resident_contigs(L, 971, f, n) for f, n in shard.window_partition(L, 8, 971)]
    assert sorted({c for h in held for c in h}) == list(range(25)) and all(2 <= len(h) <= 7 for h in held)
    assert max(sum(L[c] for c in h) for h in held) < 0.3 * sum(L)
    assert all(h == sorted(h) and h == list(range(h[0], h[-1] + 1)) for h in held)   # consecutive contigs


def test_reference_with_absent_contigs():
    """calitas_set_reference with bases[i] == NULL: contig i keeps its name and length -- the window table, windowIterator's sequence
    and the coordinates are the whole dictionary's -- but holds no bases here (a process of a multi-GPU job and the contigs its window
    range does not touch).  Host-only context: the packed size shrinks to the resident contigs, the windows of a resident contig are
    what they are in the full reference, an absent contig's bases cannot be fetched... and an index is not written from it."""
    import numpy as np
    import calitas_amd as C
    rng = np.random.default_rng(3)
    names = ["a", "b", "c"]
    seqs = [rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n, p=[0.24, 0.24, 0.24, 0.24, 0.04]) for n in (70000, 90000, 50000)]
    full = C.Context(-1); full.set_reference(names, seqs)
    part = C.Context(-1); part.set_reference(names, [None, seqs[1], None], lengths=[len(s) for s in seqs])
    try:
        assert part.contig_names == names and part.reference_info()["n_contigs"] == 3
        assert part.reference_info()["total_bases"] == full.reference_info()["total_bases"]
        assert part.reference_info()["packed_bytes"] < 0.6 * full.reference_info()["packed_bytes"]
        assert part.window_table(1000, 971, 23, chrom_index=1) == full.window_table(1000, 971, 23, chrom_index=1)
        assert part.fetch_bases(1, 500, 40) == full.fetch_bases(1, 500, 40)
        with pytest.raises(C.CalitasError):
            part.save_index("/tmp/should_not_exist.idx")
    finally:
        full.close(); part.close()


def test_release_parked_is_harmless_and_repeatable():
    """calitas_release_parked: lets go of what calitas_free has parked (the texts' blocks); with nothing parked -- and twice in a row --
    it does nothing."""
    import calitas_amd as C
    C._lib.lib.calitas_release_parked()
    ctx = C.Context(-1)
    try:
        rows = b"chr1\t1\tx\n" * 4
        import ctypes
        out = ctypes.create_string_buffer(4 * (len(b"h\t") + len(b"t\n") - 1) + len(rows) + 8)
        written = ctypes.c_uint64()
        assert C._lib.lib.calitas_expand_rows(ctx._h, rows, len(rows), 4, b"h\t", b"t\n", out, len(out), ctypes.byref(written)) == 0
    finally:
        ctx.close()
    C._lib.lib.calitas_release_parked()
    C._lib.lib.calitas_release_parked()
