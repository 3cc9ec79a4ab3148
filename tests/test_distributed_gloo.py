"""world_size-2 gloo test (CPU) of the multi-GPU partition logic in calitas_amd/shard.py: contigs are LPT-packed over
ranks, every rank produces the rows of its own contigs, rank 0 gathers the per-contig blocks over gloo and concatenates
them in dictionary order.  The compute on each rank is the CPU oracle here (this is a test; on the GPU box the same
shard/gather code wraps the HIP path in bench.py)."""
import os
import sys
import tempfile

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GUIDE = "CTTGCCCCACAGGGCAGTAAnrg"


def _genome():
    sys.path.insert(0, ROOT)
    from calitas_amd import synth
    spec = [("chr1", 26000), ("chr2", 9000), ("chr3", 15000), ("chr4", 4000), ("chr5", 6000)]
    names, seqs = synth.make_genome(spec, seed=21, guides=[("CTTGCCCCACAGGGCAGTAA", "nrg", False)], sites_per_guide=30,
                                    n_run_ends=120, n_block=900)
    return names, [s.tobytes() for s in seqs]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from calitas_amd import shard
    names, seqs = _genome()
    mine = shard.lpt_partition([len(s) for s in seqs], world)[rank]
    _, rows, _ = O.search_memory([names[i] for i in mine], [seqs[i] for i in mine], GUIDE, "a", d=4, p=1, g=2)
    if rows:
        text = "\n".join(["\t".join(rows[0].keys())] + ["\t".join(r.values()) for r in rows]) + "\n"
    else:
        text = "chromosome\n"
    hdr, blocks = shard.split_rows_by_contig(text, names)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((hdr, blocks), gathered, dst=0)
    if rank == 0:
        hdr0 = next(h for h, b in gathered if b)
        merged = shard.merge_contig_rows(hdr0, [b for _, b in gathered])
        with open(os.path.join(outdir, "merged.txt"), "w") as f:
            f.write(merged)
    dist.barrier()
    dist.destroy_process_group()


def test_contig_partition_gather_equals_single_process():
    sys.path.insert(0, HERE)
    import oracle_lib as O
    from calitas_amd import shard
    names, seqs = _genome()
    parts = shard.lpt_partition([len(s) for s in seqs], 2)
    assert sorted(parts[0] + parts[1]) == list(range(len(seqs)))
    loads = [sum(len(seqs[i]) for i in p) for p in parts]
    assert max(loads) <= 0.6 * sum(loads)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + os.getpid() % 2000
        mp.start_processes(_worker, args=(2, port, d), nprocs=2, join=True, start_method="spawn")
        merged = open(os.path.join(d, "merged.txt")).read()
    _, want, _ = O.search_memory(names, seqs, GUIDE, "a", d=4, p=1, g=2)
    lines = merged.splitlines()
    got = [dict(zip(lines[0].split("\t"), ln.split("\t"))) for ln in lines[1:]]
    assert len(want) > 5 and got == want


def _worker_contiguous(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from calitas_amd import shard
    names, seqs = _genome()
    mine = shard.contiguous_partition([len(s) for s in seqs], world)[rank]
    hdr, rows, _ = O.search_memory([names[i] for i in mine], [seqs[i] for i in mine], GUIDE, "a", d=4, p=1, g=2)
    text = ("\n".join(["\t".join(hdr)] + ["\t".join(r.values()) for r in rows]) + "\n").encode()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(text, gathered, dst=0)
    if rank == 0:
        with open(os.path.join(outdir, "whole.txt"), "wb") as f:
            f.write(shard.concat_rank_texts(gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_contiguous_partition_gather_is_a_concatenation():
    """bench.py --shard contigs: every rank owns a consecutive contig range, rank 0 concatenates the ranks' texts."""
    sys.path.insert(0, HERE)
    import oracle_lib as O
    from calitas_amd import shard, synth
    names, seqs = _genome()
    parts = shard.contiguous_partition([len(s) for s in seqs], 2)
    assert parts[0] + parts[1] == list(range(len(seqs))) and parts[0] and parts[1]
    with tempfile.TemporaryDirectory() as d:
        port = 31500 + os.getpid() % 2000
        mp.start_processes(_worker_contiguous, args=(2, port, d), nprocs=2, join=True, start_method="spawn")
        whole = open(os.path.join(d, "whole.txt")).read()
    _, want, _ = O.search_memory(names, seqs, GUIDE, "a", d=4, p=1, g=2)
    lines = whole.splitlines()
    got = [dict(zip(lines[0].split("\t"), ln.split("\t"))) for ln in lines[1:]]
    assert len(want) > 5 and got == want
    # balance on the real thing: 8 consecutive ranges of hg38's chromosomes
    for n in (2, 4, 8):
        p = shard.contiguous_partition(synth.HG38_LENGTHS, n)
        assert [i for r in p for i in r] == list(range(len(synth.HG38_LENGTHS)))
        loads = [sum(synth.HG38_LENGTHS[i] for i in r) for r in p]
        assert max(loads) <= 1.2 * sum(loads) / n, (n, loads)
    assert shard.contiguous_partition([5, 5], 4) == [[0], [1], [], []]


def test_guide_partition_is_a_partition():
    from calitas_amd import shard
    for world in (1, 2, 4, 8):
        seen = sorted(g for r in range(world) for g in shard.guides_for_rank(96, r, world))
        assert seen == list(range(96))
        sizes = [len(shard.guides_for_rank(96, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_lpt_partition_hg38_balance():
    from calitas_amd import shard, synth
    for n in (2, 4, 8):
        parts = shard.lpt_partition(synth.HG38_LENGTHS, n)
        loads = [sum(synth.HG38_LENGTHS[i] for i in p) for p in parts]
        assert max(loads) / (sum(loads) / n) < 1.06


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the script starts two fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, 127.0.0.1), they rendezvous over gloo (--dry-run: no GPU work), and rank 0's line reports n_gpus = 2.  Under a
    launcher's WORLD_SIZE that disagrees with --gpus the script refuses to run."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    threads = line.pop("worker_threads")                   # per rank: what its CPU list and its share of the cgroup's CPU quota allow
    assert line == {"dry_run": True, "n_gpus": 2, "rank_sum": 3, "local_rank": 0} and len(threads) == 2 and all(2 <= t <= 16 for t in threads)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, timeout=120)
    assert bad.returncode != 0 and b"WORLD_SIZE=2" in bad.stderr


def test_bench_launcher_stops_the_job_when_a_rank_dies():
    """A rank that dies at start-up must not leave the others waiting in the rendezvous until someone's time limit: the launcher
    watches all children, the first non-zero exit stops the rest, and the job returns that rank's code with its stderr."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--dry-run-fail-rank", "1"], env=env,
                         capture_output=True, timeout=120)
    dt = time.time() - t0
    assert out.returncode == 3, (out.returncode, out.stderr.decode()[-2000:])
    assert b"rank 1 exited with code 3" in out.stderr and b"fails on request" in out.stderr
    assert not any(ln.startswith(b"{") for ln in out.stdout.splitlines())
    assert dt < 60, dt        # rank 0 sits in the gloo rendezvous waiting for rank 1: killed by the launcher, not by a time-out


def _worker_windows(rank, world, port, outdir):
    """Window-range partition: this rank aligns the windows of its range (the oracle's per-window SequentialGuideAligner.align as
    compute), the alignments of every contig travel to the rank that owns it, and that rank runs the product's removeOverlaps /
    sort / rows (calitas_hits_tsv on a host-only context)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import calitas_amd as C
    from calitas_amd import shard
    from test_host_logic import _oracle_alignments
    names, seqs = _genome()
    lengths = [len(s) for s in seqs]
    kw = dict(d=4, p=1, g=2, D=7, O=10)
    G = C.Guide(GUIDE)
    step = 1000 - (G.cli_length + kw["d"] + kw["g"] - 1)
    ranges = shard.window_partition(lengths, world, step)
    parts = [shard.range_contigs(lengths, step, f, n) for f, n in ranges]
    owner = shard.contig_owner(parts)
    mine = {}
    for ci, k0, n, whole in parts[rank]:
        alns = _oracle_alignments(C, GUIDE, names[ci], ci, seqs[ci].decode(), kw, window_range=(k0, k0 + n))
        mine[ci] = [(a.guide_index, a.contig_index, a.window_start, a.start_offset, a.end_offset, a.guide_start_offset, a.guide_end_offset,
                     a.score, a.strand, a.pam_index, a.ops) for a in alns]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    rows = {}
    ctx = C.Context(-1)
    ctx.set_reference(names, seqs)
    params = C.make_params(max_guide_diffs=4, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2, max_total_diffs=7)
    for ci in sorted(c for c, r in owner.items() if r == rank):
        alns = []
        for r in range(world):                                   # range order = window order on the contig
            for t in gathered[r].get(ci, []):
                a = C.Alignment.__new__(C.Alignment)
                (a.guide_index, a.contig_index, a.window_start, a.start_offset, a.end_offset, a.guide_start_offset, a.guide_end_offset,
                 a.score, a.strand, a.pam_index, a.ops) = t
                alns.append(a)
        text, _ = ctx.hits_tsv(G, "a", params, alns)
        rows[ci] = text.splitlines()
    ctx.close()
    blocks = [None] * world if rank == 0 else None
    dist.gather_object(rows, blocks, dst=0)
    if rank == 0:
        merged, header = {}, None
        for b in blocks:
            for ci, lines in b.items():
                header = lines[0]
                merged[ci] = lines[1:]
        with open(os.path.join(outdir, "windows.txt"), "w") as f:
            f.write("\n".join([header] + [ln for ci in sorted(merged) for ln in merged[ci]]) + "\n")
    dist.barrier()
    dist.destroy_process_group()


def test_window_partition_cuts_a_contig_and_reproduces_the_rows():
    """bench.py --shard windows: consecutive window ranges of equal size; with two ranks the cut falls inside chr2.  The rows after the
    cross-rank merge of the cut contig equal the single-process rows, and on hg38 the ranges balance to well under 1.02."""
    sys.path.insert(0, HERE)
    import oracle_lib as O
    import calitas_amd as C
    from calitas_amd import shard, synth
    names, seqs = _genome()
    lengths = [len(s) for s in seqs]
    step = 1000 - (C.Guide(GUIDE).cli_length + 4 + 2 - 1)
    ranges = shard.window_partition(lengths, 2, step)
    parts = [shard.range_contigs(lengths, step, f, n) for f, n in ranges]
    assert sum(n for _, n in ranges) == sum(shard.window_counts(lengths, step))
    cut = [ci for ci, _, _, whole in parts[0] if not whole]
    assert cut and cut == [ci for ci, _, _, whole in parts[1] if not whole][:1]      # one contig is shared by the two ranks
    with tempfile.TemporaryDirectory() as d:
        port = 33500 + os.getpid() % 2000
        mp.start_processes(_worker_windows, args=(2, port, d), nprocs=2, join=True, start_method="spawn")
        whole = open(os.path.join(d, "windows.txt")).read()
    _, want, _ = O.search_memory(names, seqs, GUIDE, "a", d=4, p=1, g=2, D=7)
    lines = whole.splitlines()
    skip = {"aligner_version", "time_stamp", "genome_build"}
    got = [{k: v for k, v in zip(lines[0].split("\t"), ln.split("\t")) if k not in skip} for ln in lines[1:]]
    assert len(want) > 5 and got == [{k: v for k, v in r.items() if k not in skip} for r in want]
    # balance on hg38: 971-base steps, 1000-base windows
    for n in (2, 4, 8):
        loads = [shard.range_bases(synth.HG38_LENGTHS, 971, 1000, f, k) for f, k in shard.window_partition(synth.HG38_LENGTHS, n, 971)]
        assert max(loads) <= 1.02 * sum(loads) / n, (n, loads)
        assert max(loads) <= 1.0005 * sum(loads) / n


def _worker_owned(rank, world, port, outdir):
    """The partition bench.py --shard windows runs on the GPUs: every rank owns a stretch of the genome (shard.owned_stretch of its
    window range) and returns the rows whose coordinate_start lies in it; here the oracle computes the rows of the contigs a rank
    touches (the GPU path decides them from the stretch plus a halo, tests/test_gpu_binned.py holds that against this definition),
    the ranks' pieces are gathered over gloo and concatenated in rank order."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import calitas_amd as C
    from calitas_amd import shard
    from fasta_util import write_fasta
    names, seqs = _genome()
    lengths = [len(s) for s in seqs]
    step = 1000 - (C.Guide(GUIDE).cli_length + 4 + 2 - 1)
    first, n = shard.window_partition(lengths, world, step)[rank]
    stretch = shard.owned_stretch(lengths, step, first, n)
    touched = sorted({ci for ci, _, _, _ in shard.range_contigs(lengths, step, first, n)})
    fa = write_fasta(os.path.join(outdir, "owned_%d.fa" % rank), [(names[ci], seqs[ci].decode()) for ci in touched])
    _, rows, _ = O.search_reference(fa, GUIDE, "a", d=4, p=1, g=2, D=7, threads=2)
    mine = [r for r in rows if shard.owns(stretch, names.index(r["chromosome"]), int(r["coordinate_start"]))]
    pieces = [None] * world if rank == 0 else None
    dist.gather_object(mine, pieces, dst=0)
    if rank == 0:
        import json
        with open(os.path.join(outdir, "owned.json"), "w") as f:
            json.dump([r for piece in pieces for r in piece], f)
    dist.barrier()
    dist.destroy_process_group()


def test_owned_stretches_concatenate_to_the_whole_job():
    """world_size 2 over gloo: the window partition cuts a contig; each rank keeps the rows its stretch owns; the concatenation in rank
    order is the single-process hits.txt, row for row and in order.  And the stretches of hg38's 8-way partition tile the genome."""
    sys.path.insert(0, HERE)
    import json
    import oracle_lib as O
    import calitas_amd as C
    from calitas_amd import shard, synth
    from fasta_util import write_fasta
    names, seqs = _genome()
    lengths = [len(s) for s in seqs]
    step = 1000 - (C.Guide(GUIDE).cli_length + 4 + 2 - 1)
    with tempfile.TemporaryDirectory() as d:
        port = 33500 + os.getpid() % 2000
        mp.start_processes(_worker_owned, args=(2, port, d), nprocs=2, join=True, start_method="spawn")
        got = json.load(open(os.path.join(d, "owned.json")))
        fa = write_fasta(os.path.join(d, "whole.fa"), [(n, s.decode()) for n, s in zip(names, seqs)])
        _, want, _ = O.search_reference(fa, GUIDE, "a", d=4, p=1, g=2, D=7, threads=2)
    skip = {"aligner_version", "time_stamp"}
    strip = lambda rows: [{k: v for k, v in r.items() if k not in skip} for r in rows]
    assert len(want) > 5 and strip(got) == strip(want)
    # hg38, 8 ranks: the stretches are consecutive, start at (0, 0), end at the end of the reference, and every cut is a window start
    hstep = 971
    parts = shard.window_partition(synth.HG38_LENGTHS, 8, hstep)
    st = [shard.owned_stretch(synth.HG38_LENGTHS, hstep, f, n) for f, n in parts]
    assert st[0][0] == (0, 0) and st[-1][1] == (len(synth.HG38_LENGTHS), 0)
    for a, b in zip(st[:-1], st[1:]):
        assert a[1] == b[0] and a[1][1] % hstep == 0
    loads = [shard.range_bases(synth.HG38_LENGTHS, hstep, 1000, f, n) for f, n in parts]
    assert max(loads) / (sum(loads) / 8) < 1.001


GUIDES3 = [GUIDE, "GTGACTTGAAGTCTCAGTATnrg", "ACGTTGCATGCATGCCATGAnrg"]


def _worker_batch_sharded(rank, world, port, outdir):
    """bench.py's batch_sharded block with the oracle as compute: every rank runs ALL guides on its stretch (the rows it owns per guide),
    reports (crc32, bytes, rows) of each text over a gloo all_gather, and rank 0 -- which has the whole genome -- holds every rank's text
    against its consecutive piece of the single-process text (shard.pieces_match, the function bench.py calls)."""
    import zlib
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import calitas_amd as C
    from calitas_amd import shard
    from fasta_util import write_fasta
    names, seqs = _genome()
    lengths = [len(s) for s in seqs]
    step = 1000 - (C.Guide(GUIDE).cli_length + 4 + 2 - 1)
    first, n = shard.window_partition(lengths, world, step)[rank]
    stretch = shard.owned_stretch(lengths, step, first, n)
    touched = sorted({ci for ci, _, _, _ in shard.range_contigs(lengths, step, first, n)})
    fa = write_fasta(os.path.join(outdir, "bs_%d.fa" % rank), [(names[ci], seqs[ci].decode()) for ci in touched])
    fa_all = write_fasta(os.path.join(outdir, "bs_all_%d.fa" % rank), [(nm, s.decode()) for nm, s in zip(names, seqs)])

    def text_of(header, rows):
        return ("\t".join(header) + "\n" + "".join("\t".join(r[h] for h in header) + "\n" for r in rows)).encode()
    stats = []
    for gi, g in enumerate(GUIDES3):
        header, rows, _ = O.search_reference(fa, g, "g%d" % gi, d=4, p=1, g=2, D=7, threads=2)
        mine = [r for r in rows if shard.owns(stretch, names.index(r["chromosome"]), int(r["coordinate_start"]))]
        for r in mine + rows:
            r["time_stamp"] = "t"; r["aligner_version"] = "v"            # (run-dependent columns: fixed, as the bench passes them)
        t = text_of(header, mine)
        stats += [float(zlib.crc32(t)), float(len(t)), float(len(mine))]
    mine_t = torch.tensor(stats, dtype=torch.float64)
    gathered = [torch.zeros_like(mine_t) for _ in range(world)]
    dist.all_gather(gathered, mine_t)
    if rank == 0:
        ok, tampered = [], []
        for gi, g in enumerate(GUIDES3):
            header, rows, _ = O.search_reference(fa_all, g, "g%d" % gi, d=4, p=1, g=2, D=7, threads=2)
            for r in rows:
                r["time_stamp"] = "t"; r["aligner_version"] = "v"
            whole = text_of(header, rows)
            pieces = [tuple(int(x) for x in gathered[r][3 * gi:3 * gi + 3].tolist()) for r in range(world)]
            ok.append(shard.pieces_match(whole, pieces, len(rows)))
            bad = [pieces[0]] + [(c ^ 1, nb, nr) for c, nb, nr in pieces[1:]]
            tampered.append(shard.pieces_match(whole, bad, len(rows)) or shard.pieces_match(whole, pieces, len(rows) + 1)
                            or shard.pieces_match(whole + b"x\n", pieces, len(rows)))
        import json
        json.dump({"ok": ok, "tampered": tampered, "rows": [int(gathered[r][2]) for r in range(world)]}, open(os.path.join(outdir, "bs.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharded_check_over_gloo():
    """world_size 2: the check of bench.py's batch_sharded block -- every rank all guides on its stretch, (crc, bytes, rows) gathered,
    rank 0 compares with single-process texts -- accepts the true partition and refuses a wrong CRC, a wrong row count and a longer text."""
    import json
    with tempfile.TemporaryDirectory() as d:
        port = 35600 + os.getpid() % 2000
        mp.start_processes(_worker_batch_sharded, args=(2, port, d), nprocs=2, join=True, start_method="spawn")
        res = json.load(open(os.path.join(d, "bs.json")))
    assert res["ok"] == [True, True, True] and res["tampered"] == [False, False, False] and all(r >= 0 for r in res["rows"])


def test_bench_ranks_pin_themselves_to_their_gpus_numa_node(tmp_path):
    """`python bench.py --gpus 2 --dry-run` with a KFD topology to read (a fake /sys: two GPUs on one NUMA node whose CPUs are the ones this
    process may use): every rank ends up on its half of the node's CPUs before it starts a thread; without a topology nothing is touched."""
    import json
    import subprocess
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("fewer than 4 CPUs")
    def put(rel, text):
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    put("sys/class/kfd/kfd/topology/nodes/0/properties", "cpu_cores_count %d\nsimd_count 0\ndrm_render_minor -1\n" % len(allowed))
    for g in range(2):
        put("sys/class/kfd/kfd/topology/nodes/%d/properties" % (1 + g), "cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % (128 + g))
        put("sys/class/drm/renderD%d/device/numa_node" % (128 + g), "0\n")
    put("sys/devices/system/node/node0/cpulist", ",".join(str(c) for c in allowed) + "\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "CALITAS_THREADS")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, CALITAS_BENCH_SYSFS_ROOT=str(tmp_path)),
                         capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    half = len(allowed) // 2
    assert line["rank_cpus"] == [allowed[:half], allowed[half:2 * half]]
    assert line["worker_threads"] == [max(2, min(16, half))] * 2          # no CPU quota in the fake tree: what the CPU list allows
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, CALITAS_BENCH_SYSFS_ROOT=str(tmp_path / "nowhere")),
                           capture_output=True, timeout=300)
    assert plain.returncode == 0 and "rank_cpus" not in json.loads(plain.stdout.decode().strip().splitlines()[-1])
    # a CPU quota on the way up the cgroup tree (what a GPU box has: cpu.max 600000 100000 = six cores' worth): a rank's pool gets its share of it
    put("proc/self/cgroup", "0::/kubepods/pod1/box\n")
    put("sys/fs/cgroup/kubepods/pod1/box/cpu.max", "max 100000\n")
    put("sys/fs/cgroup/kubepods/pod1/cpu.max", "600000 100000\n")
    put("sys/fs/cgroup/cpu.max", "max 100000\n")
    quota = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, CALITAS_BENCH_SYSFS_ROOT=str(tmp_path)),
                           capture_output=True, timeout=300)
    assert quota.returncode == 0, quota.stderr.decode()[-2000:]
    assert json.loads(quota.stdout.decode().strip().splitlines()[-1])["worker_threads"] == [max(2, min(half, 3))] * 2
