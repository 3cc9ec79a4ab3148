#!/usr/bin/env python3
"""bench.py -- SearchReference full-scan throughput on MI355X (BASELINE.json metric).

Default workload (--config 3, N=1): BASELINE config 3 -- one 20-nt guide + NRG PAM against a synthetic hg38-sized genome (25 contigs,
3 088 286 401 bp, N runs, soft-masking, tandem repeats, planted sites), max-guide-diffs 5, max-pam-mismatches 1,
max-gaps-between-guide-and-pam 2.  A "step" is one complete SearchReference pass for one guide over the resident reference:
scan kernel + aligner kernels + per-window filter + removeOverlaps / sort / all hits.txt rows on the device + copy-back of the text
(everything except writing hits.txt to disk).  The packed reference is resident in HBM before the timed region.

--gpus N > 1: one process per GPU.  Started by a launcher (torchrun: WORLD_SIZE / RANK / LOCAL_RANK in the environment) the script
is one rank; started plainly (`python bench.py --gpus 8`) it spawns the N ranks itself -- fresh child processes, decided before
this process touches a GPU, the first one that fails stops the others -- and prints rank 0's line.  RCCL carries the barriers around
the timed region; there is no data-path collective.  The N > 1 headline is ONE pass divided over the ranks (strong scaling):
--shard windows (default) cuts windowIterator's windows into N equal consecutive ranges, every rank returns the rows whose
coordinate_start lies in its stretch (calitas_search_hits_into on a window range) straight into its slot of one shared-memory file,
and rank 0 checks the gathered file against a single-process search byte for byte after the timed region; --shard contigs gives every
rank whole contigs (BASELINE's wording; 1.20x off balance at 8 ranks); --shard guides is the weak-scaling mode (every rank its own
guide pass over the whole genome), also measured as a labelled second figure with --secondary.

--config 4: BASELINE config 4, the 96-guide batch (guide #0 + 95 random 20-mers, seed 0xC4) through calitas_search_hits_batch;
a step = all 96 guides.  --config 5: BASELINE config 5's shape, PAM-less 20-mer, max-guide-diffs 8, with a synthetic VCF
(one variant per kilobase, seed 0xC5) through calitas_search_variants; its default --scale is reduced.

value = candidate loci examined per second = 2 strands x reference bases x guide passes / wall time.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GUIDE0 = "CTTGCCCCACAGGGCAGTAAnrg"   # README.md:74 of the reference
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# vector-ALU issue roof of the chip: 256 CUs x 4 SIMD-32, a wave64 instruction every 2 cycles at 2.4 GHz (MI355X_MICROARCH.md,
# "Wave scheduling" and the cycle-constants table), in wave-instructions per nanosecond
VALU_PEAK_GUIDE = 256 * 4 * 2.4 / 2


def gen_contig(length, seed, device, n_ends, n_block, softmask=0.5, tandem_frac=0.01, gc=0.41):
    """One synthetic contig as a numpy uint8 array of ASCII bases, generated on the GPU with torch (plumbing only)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.randint(0, 256, (length,), dtype=torch.uint8, device=device, generator=g)
    at = (1.0 - gc) / 2
    t0, t1, t2 = int(at * 256), int((at + gc / 2) * 256), int((at + gc) * 256)
    idx = (u >= t0).to(torch.uint8) + (u >= t1).to(torch.uint8) + (u >= t2).to(torch.uint8)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    seq = lut[idx.long()]
    del u, idx
    cg = torch.Generator(device="cpu")
    cg.manual_seed(seed + 1)
    if tandem_frac > 0 and length > 1000:
        n_tr = max(1, int(length * tandem_frac / 150))
        starts = torch.randint(0, max(1, length - 400), (n_tr,), generator=cg).tolist()
        units = torch.randint(2, 8, (n_tr,), generator=cg).tolist()
        tracts = torch.randint(30, 300, (n_tr,), generator=cg).tolist()
        if n_tr > 20000:   # keep the python loop bounded on the big contigs
            starts, units, tracts = starts[:20000], units[:20000], tracts[:20000]
        for s, unit, tract in zip(starts, units, tracts):
            e = min(length, s + tract)
            seq[s:e] = seq[s:s + unit].repeat(tract // unit + 1)[:e - s]
    if softmask > 0 and length > 2000:
        n_runs = int(length * softmask / 2650)
        starts = torch.randint(0, length, (n_runs,), device=device, generator=g)
        lens = torch.randint(300, 5000, (n_runs,), device=device, generator=g)
        delta = torch.zeros(length + 1, dtype=torch.int32, device=device)
        delta.index_add_(0, starts, torch.ones(n_runs, dtype=torch.int32, device=device))
        delta.index_add_(0, torch.clamp(starts + lens, max=length), -torch.ones(n_runs, dtype=torch.int32, device=device))
        low = torch.cumsum(delta[:-1], 0) > 0
        seq = torch.where(low, seq | 0x20, seq)
        del delta, low
    if n_ends > 0:
        seq[:min(n_ends, length)] = ord("N")
        seq[max(0, length - n_ends):] = ord("N")
    if n_block > 0 and length > 3 * n_block:
        s = int(torch.randint(length // 3, 2 * length // 3 - n_block, (1,), generator=cg))
        seq[s:s + n_block] = ord("N")
    out = seq.cpu().numpy()
    del seq
    return out


def build_genome(scale, device, contig_indices=None, guides=(), log=None):
    """hg38-sized synthetic genome (seed 0xC3). Returns (names, arrays) for the requested contigs."""
    import numpy as np
    from calitas_amd import synth
    spec = synth.hg38_like_spec(scale)
    names, seqs = [], []
    for ci, (name, length) in enumerate(spec):
        if contig_indices is not None and ci not in contig_indices:
            continue
        big = length > 2_000_000
        rng = np.random.default_rng([0xC3, ci])
        n_block = int(rng.integers(1_000_000, 3_000_000) * min(1.0, scale * 4)) if big else 0
        seq = gen_contig(length, 0xC300 + ci, device, n_ends=10_000 if big else 0, n_block=n_block)
        # planted sites for each guide: 0-6 edits, both strands, some straddling window starts (k*971)
        n_sites = max(2, int(40 * length / 3.1e9))
        for gi, gstr in enumerate(guides):
            proto, pam = gstr[:20], gstr[20:]
            for k in range(n_sites):
                lo = 10_000 if big else 0
                pos = int(rng.integers(lo + 100, max(lo + 200, length - lo - 100)))
                if k % 3 == 0:
                    pos = (pos // 971) * 971 - int(rng.integers(0, 30))
                synth.plant_site(rng, seq, pos, proto, pam, False, int(rng.integers(0, 6)), bool(rng.integers(0, 2)))
        names.append(name)
        seqs.append(seq)
        if log:
            log("generated %s (%d bp)" % (name, length))
    return names, seqs


def synthetic_vcf(path, names, seqs, per_kb=1.0, seed=0xC5):
    """Biallelic SNVs and short indels at about one per kilobase, AF in [0.01, 0.5] (BASELINE config 5)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    n_var = 0
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele frequency\">\n")
        for nm, s in zip(names, seqs):
            f.write("##contig=<ID=%s,length=%d>\n" % (nm, len(s)))
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for nm, s in zip(names, seqs):
            n = int(len(s) * per_kb / 1000)
            if n == 0:
                continue
            pos = np.unique(rng.integers(50, max(51, len(s) - 50), size=n))
            pos = pos[np.concatenate(([True], np.diff(pos) > 8))]          # keep REF spans apart
            kinds = rng.integers(0, 4, size=len(pos))
            afs = rng.uniform(0.01, 0.5, size=len(pos))
            alt_pick = rng.integers(0, 3, size=len(pos))
            ins = rng.integers(0, 4, size=(len(pos), 3))
            lines = []
            for p, k, af, ap, iv in zip(pos.tolist(), kinds.tolist(), afs.tolist(), alt_pick.tolist(), ins.tolist()):
                ref = chr(s[p - 1] & 0xDF)
                if ref not in "ACGT":
                    continue
                if k <= 1:
                    alt = [b for b in "ACGT" if b != ref][ap]
                elif k == 2:
                    alt = ref + "".join("ACGT"[x] for x in iv[:1 + ap])
                else:
                    span = bytes(s[p - 1:p + 1 + ap]).decode().upper()
                    if any(c not in "ACGT" for c in span):
                        continue
                    ref, alt = span, span[0]
                lines.append("%s\t%d\trs%d\t%s\t%s\t.\t.\tAF=%.3f\n" % (nm, p, n_var, ref, alt, af))
                n_var += 1
            f.write("".join(lines))
    return n_var


def host_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(names, seqs, guides, params_kw, budget_bases, guide_ids=None):
    """Times the CPU oracle (the restatement of the reference algorithm, oracle/) on a bounded sample of the same genome with one
    worker per host core -- the reference's own threading model (SearchReference.scala:459).  Returns (report, {contig: rows}) with
    the oracle's rows of the whole contigs of the sample (for the parity check of the bench's own output)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = min(host_cores(), 16)   # the GPU box gives one GPU a 16-core CPU share
    # sample: whole contigs of the same genome, in order, until the budget is reached (the last one truncated)
    s_names, s_seqs, total, whole = [], [], 0, []
    for n, s in zip(names, seqs):
        if total >= budget_bases:
            break
        take = min(len(s), budget_bases - total)
        s_names.append(n); s_seqs.append(bytes(s[:take])); total += take
        if take == len(s):
            whole.append(n)
    t0 = time.perf_counter()
    rows_by_guide, nwin = [], 0
    for gi, g in enumerate(guides):
        _, rows, nwin = O.search_memory(s_names, s_seqs, g, guide_ids[gi] if guide_ids else "bench", d=params_kw["max_guide_diffs"], p=params_kw["max_pam_mismatches"],
                                        g=params_kw["max_gaps_between_guide_and_pam"], threads=cores)
        rows_by_guide.append(rows)
    dt = time.perf_counter() - t0
    report = {"value": 2 * total * len(guides) / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
              "sample": "%d bp (%s%s; %d windows, %d guide pass(es), %d hits) in %.1f s; oracle/ C++ restatement of the reference algorithm, %d threads"
                        % (total, ",".join(s_names[:3]), "..." if len(s_names) > 3 else "", nwin, len(guides), sum(len(r) for r in rows_by_guide), dt, cores),
              "bases_per_s": total * len(guides) / dt}
    return report, whole, rows_by_guide


def cpu_baseline_c5(names, seqs, guide, params_kw, budget_bases, parity_bases=16_000_000):
    """BASELINE config 5's CPU leg: the oracle's SearchReference --variants (search_reference_vcf) on a prefix of the first contig with
    a synthetic VCF of the same density, one worker per host core; and, on a shorter prefix, the GPU path (calitas_search_variants on a
    context of its own) against the oracle's rows -- as a MULTISET: rows whose sort keys tie between a variant group and the reference
    group come in a hash-map order in the reference (SearchReference.scala:656), neither side pins it.  Returns (report, parity)."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from fasta_util import write_fasta
    import calitas_amd as C
    from calitas_amd import _lib
    cores = min(host_cores(), 16)
    kw = dict(d=params_kw["max_guide_diffs"], p=params_kw["max_pam_mismatches"], g=params_kw["max_gaps_between_guide_and_pam"])
    tmp = "/dev/shm/calitas_bench_c5cpu_%d" % os.getpid()
    os.makedirs(tmp, exist_ok=True)
    made = []

    def sample(n_bases, tag):
        take = min(len(seqs[0]), n_bases)
        seq = bytes(seqs[0][:take])
        fa = write_fasta(os.path.join(tmp, tag + ".fa"), [(names[0], seq.decode())])
        vcf = os.path.join(tmp, tag + ".vcf")
        n_var = synthetic_vcf(vcf, [names[0]], [seq])
        made.extend([fa, fa + ".fai", os.path.splitext(fa)[0] + ".dict", vcf])
        return fa, vcf, seq, n_var, take

    try:
        fa, vcf, _, n_var, take = sample(budget_bases, "time")
        t0 = time.perf_counter()
        _, rows, nwin = O.search_reference_vcf(fa, vcf, guide, "bench", threads=cores, **kw)
        dt = time.perf_counter() - t0
        report = {"value": 2 * take / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
                  "sample": "%d bp of %s with a synthetic VCF (%d variants; %d windows, %d hits) in %.1f s; oracle/ C++ restatement of "
                            "SearchReference --variants, %d threads" % (take, names[0], n_var, nwin, len(rows), dt, cores),
                  "bases_per_s": take / dt}
        del rows
        # parity on a shorter prefix
        fa2, vcf2, seq2, n_var2, take2 = sample(parity_bases, "parity")
        _, want, _ = O.search_reference_vcf(fa2, vcf2, guide, "bench", threads=cores, **kw)
        ctx = C.Context(0)
        try:
            ctx.set_reference([names[0]], [seq2], genome_build="testassembly")
            g = C.Guide(guide).to_c()
            params = C.make_params(**params_kw)
            tsv, nbytes, nrows, nw = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
            _lib.check(ctx._h, _lib.lib.calitas_search_variants(ctx._h, ctypes.byref(g), b"bench", ctypes.byref(params), vcf2.encode(), None, None, b"v", b"t",
                                                              ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(nrows), ctypes.byref(nw)))
            text = ctypes.string_at(tsv.value, nbytes.value).decode()
            _lib.lib.calitas_free(tsv)
        finally:
            ctx.close()
        skip = {"aligner_version", "time_stamp"}

        def norm(rs):
            out = []
            for r in rs:
                r = {k: v for k, v in r.items() if k not in skip}
                if r.get("variant_vcf"):
                    r["variant_vcf"] = r["variant_vcf"].split(":")[0]            # the oracle leaves the md5 out
                out.append(json.dumps(r, sort_keys=True))
            return sorted(out)
        got = C.read_hits(text)
        same = norm(got) == norm(want)
        parity = {"contigs": ["%s[:%d]" % (names[0], take2)], "guides": 1, "rows": len(want), "variants": n_var2, "identical": bool(same),
                  "compared_as": "multiset (rows whose sort keys tie between a variant group and the reference group have no pinned order, SR:656)",
                  "rows_with_a_variant": sum(1 for r in got if r.get("variant_id"))}
        return report, parity
    finally:
        for f in made:
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes (this one has not imported torch or touched a GPU),
    rank 0's JSON line passed through.  A failing rank fails the run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    import threading
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        ef = tempfile.TemporaryFile()                       # every rank's stderr: shown when that rank fails
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef))
    # rank 0's stdout is drained on a thread, so that this loop can watch ALL children: the first one that fails ends the job (the
    # others would sit in the rendezvous or a barrier until somebody's time limit)
    out0 = []
    drain = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    drain.start()
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
        elif all(c == 0 for c in codes):
            break
        else:
            time.sleep(0.05)
    if failed is not None:
        for p in procs:                                     # fresh children of this process: kill, never re-exec
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        errs[failed].seek(0)
        tail = errs[failed].read().decode(errors="replace")[-4000:]
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were stopped.  Its stderr (tail):\n%s\n"
                         % (failed, procs[failed].returncode, tail))
        raise SystemExit(procs[failed].returncode if 0 < procs[failed].returncode < 256 else 1)
    drain.join()
    errs[0].seek(0)
    sys.stderr.write(errs[0].read().decode(errors="replace"))     # rank 0's log lines
    out0 = (out0[0] if out0 else b"").decode()
    line = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not line:
        raise SystemExit("bench.py: rank 0 printed no result line")
    if json.loads(line[-1]).get("n_gpus") != n:
        raise SystemExit("bench.py: the result line does not report n_gpus = %d" % n)
    print(line[-1], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--prime-seconds", type=float, default=None,
                    help="untimed calls before the warm-up steps until the device runs at its sustained clocks (default 0.75 s for config 3, 0 otherwise)")
    ap.add_argument("--config", type=int, choices=[3, 4, 5], default=3, help="BASELINE config: 3 one guide (the metric's), 4 the 96-guide batch, 5 PAM-less d=8 + VCF")
    ap.add_argument("--scale", type=float, default=None, help="genome size relative to hg38 (1.0 = 3.09 Gb; default 1.0)")
    ap.add_argument("--shard", choices=["contigs", "windows", "guides"], default="windows",
                    help="N>1: windows (default) = consecutive window ranges of ONE pass, equal to within a window, contigs cut where the balance asks "
                         "for it (strong scaling; every rank owns the rows of its stretch); contigs = consecutive whole-contig ranges of one pass "
                         "(BASELINE's wording; 1.20x off balance at 8 ranks); guides = every rank its own guide pass over the whole genome (weak)")
    ap.add_argument("--cpu-sample-mb", type=float, default=-1, help="CPU baseline sample in Mb (<0: auto, 0: skip)")
    ap.add_argument("--secondary", action="store_true",
                    help="N>1: also measure the partition mode that is not the headline (every rank regenerates and re-uploads the genome for it)")
    ap.add_argument("--no-secondary", action="store_true", help=argparse.SUPPRESS)   # the default since round 3; accepted for old command lines
    ap.add_argument("--batch-sharded", dest="batch_sharded", action="store_true", default=None,
                    help="also measure BASELINE config 4's shape on this partition: every rank runs all 96 guides on its window range "
                         "(calitas_search_hits_batch on a range), a sample of guides checked against single-process calls; default for N > 1")
    ap.add_argument("--no-batch-sharded", dest="batch_sharded", action="store_false")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)   # tests: this rank exits 3 before the rendezvous
    ap.add_argument("--text-block", choices=["caller", "library"], default="caller",
                    help="--config 5: the text goes to a page-locked buffer of the bench (calitas_search_variants_into) or to a block of the library's per call")
    ap.add_argument("--no-hits", action="store_true", help="time the search only (no removeOverlaps / row building)")
    ap.add_argument("--guides-per-step", type=int, default=1,
                    help="config 3: guides each rank runs per step through calitas_search_hits_batch; the default 1 is the BASELINE metric's single-guide pass")
    ap.add_argument("--same-guide", action="store_true", help="with --guides-per-step: every guide of the batch is guide #0 (isolates the pipelining gain)")
    ap.add_argument("--distinct-guides", action="store_true", help="--shard guides: rank r runs guide #r of the 96-guide set instead of guide #0 on every rank")
    ap.add_argument("--two-stage", action="store_true", help="calitas_search + calitas_hits_tsv (host rows) instead of the fused calitas_search_hits")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and gloo replaces RCCL (not a measurement)")
    ap.add_argument("--dry-run", action="store_true",
                    help="the ranks rendezvous over gloo, reduce one number and rank 0 prints a stub line: exercises the launch path without a GPU")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)          # before torch is imported: the children are the first to touch a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or with torchrun --nproc-per-node N and --gpus N)"
                         % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank_info = {}
    def pin_this_rank():
        """N > 1: every rank on the CPUs of its GPU's NUMA node (a rank's slice is bound by host round trips, and one from the far socket
        costs a quarter more: DESIGN.md 4.7), its share of them when several GPUs hang off one node -- before anything starts a thread.
        (calitas_amd/shard.py loaded by path: the package itself loads the HIP library, which has to come after torch.)"""
        if world <= 1:
            return None
        import importlib.util
        spec = importlib.util.spec_from_file_location("_calitas_shard", os.path.join(ROOT, "calitas_amd", "shard.py"))
        shard_mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(shard_mod)
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        cpus = None
        if not (args.rehearse_on_one_gpu or os.environ.get("CALITAS_BENCH_PIN", "1") == "0"):
            cpus = shard_mod.pin_rank(local_rank, n_local)
        if "CALITAS_THREADS" not in os.environ:
            # the library's worker pool: a thread per CPU the rank may use, and no more than its share of the box's CPU quota (a quota
            # on a shared host, not a CPU set: eight ranks with sixteen workers each on sixteen cores' worth throttle each other)
            os.environ["CALITAS_THREADS"] = str(shard_mod.worker_threads(cpus or sorted(os.sched_getaffinity(0)), n_local, shard_mod.cgroup_cpu_quota()))
        return cpus

    if args.dry_run:
        if rank == args.dry_run_fail_rank:
            sys.stderr.write("bench.py: rank %d fails on request (--dry-run-fail-rank)\n" % rank)
            raise SystemExit(3)
        pinned = pin_this_rank()
        import torch
        import torch.distributed as dist
        total = rank + 1
        cpus = [sorted(os.sched_getaffinity(0))]
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([rank + 1], dtype=torch.int64)
            dist.all_reduce(t)
            total = int(t.item())
            cpus = [None] * world
            dist.all_gather_object(cpus, sorted(os.sched_getaffinity(0)))
            threads = [None] * world
            dist.all_gather_object(threads, int(os.environ.get("CALITAS_THREADS", "0") or 0))
            dist.barrier()
            dist.destroy_process_group()
        else:
            threads = [int(os.environ.get("CALITAS_THREADS", "0") or 0)]
        if rank == 0:
            line = {"dry_run": True, "n_gpus": world, "rank_sum": total, "local_rank": local_rank, "worker_threads": threads}
            if pinned:
                line["rank_cpus"] = cpus                          # (only when a topology was found: the affinity every rank ended up with)
            print(json.dumps(line), flush=True)
        return
    if args.steps is None:
        args.steps = {3: 20, 4: 2, 5: 2}[args.config]
    if args.warmup is None:
        args.warmup = {3: 3, 4: 1, 5: 1}[args.config]
    if args.prime_seconds is None:
        args.prime_seconds = 0.75 if args.config == 3 else 0.0
    if args.scale is None:
        args.scale = 1.0                                       # every config at its stated size (config 5 takes ~1.2 s per step since round 4)

    pinned = pin_this_rank()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    gloo = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
            gloo = dist.group.WORLD
        else:
            dist.init_process_group("nccl", device_id=device)
            gloo = dist.new_group(backend="gloo")

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    import calitas_amd as C
    from calitas_amd import shard, synth

    all_guides = [GUIDE0] + synth.random_guides(0xC4, 95)
    if args.config == 5:
        params_kw = dict(max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
        all_guides = [GUIDE0[:20]]                                         # PAM-less guide #0
        workload = "SearchReference --variants: PAM-less 20 nt guide, max-guide-diffs=8, synthetic VCF (1 variant / kb)"
    else:
        params_kw = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
        workload = ("SearchReference: 96-guide batch (20 nt + NRG PAM)" if args.config == 4 else "SearchReference: 20 nt guide + NRG PAM")
    params = C.make_params(**params_kw)
    spec = synth.hg38_like_spec(args.scale)
    lengths = [l for _, l in spec]
    workload += " vs synthetic hg38-sized genome (25 contigs, %d bp), " % sum(lengths) + " ".join(
        "%s=%d" % (k.replace("_", "-"), v) for k, v in params_kw.items())

    # ---- what this rank holds and runs ----
    def partition_mode(mode):
        """(contig indices of this rank or None = all, its guides, guide passes per step over the job, bases per step over the job)"""
        if args.config == 4:
            gl = all_guides
        elif args.config == 5:
            gl = all_guides[:1]
        else:
            gl = None
        if world > 1 and mode == "windows":
            # windowIterator's windows in N consecutive ranges of equal size, cut wherever that falls (max / mean of the bases per rank
            # 1.0000-1.0005): every rank holds the whole packed genome, scans its stretch and returns the rows whose coordinate_start
            # lies in it (calitas_search_hits_into on a window range) -- consecutive pieces of hits.txt, no exchange between the ranks
            g = [GUIDE0]
            return None, g, 1, sum(lengths)
        if world > 1 and mode == "contigs":
            mine = shard.contiguous_partition(lengths, world)[rank]        # consecutive ranges: the gather is a concatenation
            g = gl or [GUIDE0]
            return mine, g, len(g), sum(lengths) * len(g)                  # the ranks share ONE pass per guide
        gps = max(1, args.guides_per_step)
        if gl is None:
            # weak scaling keeps the work per GPU fixed: every rank runs the pass of the N=1 line (guide #0); the 96-guide set's random
            # 20-mers yield ~3x the rows of guide #0 on this genome (copy-back bound) and are drawn with --distinct-guides or a batch
            distinct = args.distinct_guides or (gps > 1 and not args.same_guide)
            gl = [all_guides[(rank * gps + i) % len(all_guides)] if distinct else GUIDE0 for i in range(gps)]
        return None, gl, world * len(gl), sum(lengths) * world * len(gl)

    def make_context(mine, resident=None):
        """mine: the contigs this rank's context consists of (None: all).  resident (window partition): all contigs are in the
        context -- windowIterator's sequence, coordinates and order are the whole genome's -- but only these have their bases here."""
        t_gen = time.perf_counter()
        names, seqs = build_genome(args.scale, device, contig_indices=mine if resident is None else resident, guides=[GUIDE0], log=None)
        log("genome: %d contigs, %d bp on this rank, generated in %.1f s" % (len(names), sum(len(s) for s in seqs), time.perf_counter() - t_gen))
        ctx = C.Context(local_rank)
        t_set = time.perf_counter()
        if resident is not None:
            have = dict(zip(names, seqs))
            names = [n for n, _ in spec]
            seqs = [have.get(n) for n in names]
            ctx.set_reference(names, seqs, genome_build="synthetic-hg38-sized", lengths=lengths)
            rank_info["resident_contigs"] = len(have); rank_info["resident_bases"] = sum(len(s) for s in have.values())
        else:
            ctx.set_reference(names, seqs, genome_build="synthetic-hg38-sized")
        log("set_reference: %.2f s (pack + upload), %d packed bytes" % (time.perf_counter() - t_set, ctx.reference_info()["packed_bytes"]))
        return ctx, names, seqs

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def measure_batch_sharded(ctx, params_rank, names, seqs):
        """BASELINE config 4 on this job's partition: every rank runs ALL 96 guides on its window range (calitas_search_hits_batch on a
        range: the rows it owns per guide, guides pipelined through the device stages), nothing exchanged; a step ends when the slowest
        rank has its 96 texts.  A sample of guides is then held against single-process calls on rank 0, which has the whole genome: a rank's
        text of guide g must be the header plus its consecutive piece of the whole text (CRC and length per rank)."""
        import zlib
        G96 = [C.Guide(g) for g in all_guides]
        ids96 = ["g%02d" % i for i in range(len(G96))]
        ctx.search_hits_batch(G96, ids96, params_rank, "bench", "bench", decode=False)          # untimed: text blocks, lane buffers
        steps_b = 2
        sync()
        t0 = time.perf_counter()
        for _ in range(steps_b):
            res = ctx.search_hits_batch(G96, ids96, params_rank, "bench", "bench", decode=False)
        sync()
        dt = time.perf_counter() - t0
        tm = ctx.timing()
        rows_rank, bytes_rank = sum(r for _, r in res), sum(b for b, _ in res)
        sample = [0, 37, 95]
        mine = ctx.search_hits_batch([G96[i] for i in sample], [ids96[i] for i in sample], params_rank, "bench", "bench", decode="digest")
        stats = [dt, float(rows_rank), float(bytes_rank)] + [float(x) for (crc, nb), rows in mine for x in (crc, nb, rows)]
        allst = [stats]
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor(stats, dtype=torch.float64)
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t, group=gloo)
            allst = [g.tolist() for g in gathered]
        if rank != 0:
            return None
        dt_max = max(a[0] for a in allst)
        check = {"guides": sample, "identical": True, "rows": 0}
        for k, gi in enumerate(sample):
            whole, n_whole = ctx.search_hits(G96[gi], ids96[gi], params, "bench", "bench", decode="bytes")
            pieces = [tuple(int(allst[r][3 + 3 * k + j]) for j in range(3)) for r in range(world)]
            if not shard.pieces_match(whole, pieces, n_whole):
                check["identical"] = False
            check["rows"] += n_whole
        total_bases = sum(lengths)
        block = {"workload": "SearchReference: 96-guide batch (BASELINE config 4), every rank all guides on its window range",
                 "guides_per_step": len(G96), "steps": steps_b, "ms_per_step": dt_max / steps_b * 1e3,
                 "value": 2.0 * total_bases * len(G96) * steps_b / dt_max, "unit": "candidates/s", "scaling": "strong",
                 "rows_per_step": int(sum(a[1] for a in allst)), "text_bytes_per_step": int(sum(a[2] for a in allst)),
                 "rank_seconds": [round(a[0], 4) for a in allst], "lanes": tm.get("lanes"), "binned_lanes_rank0": tm.get("binned_lanes"),
                 "check": check}
        log("batch_sharded: %.1f ms per 96-guide step on %d rank(s), sample of guides identical to single-process calls: %s"
            % (block["ms_per_step"], world, check["identical"]))
        return block

    def measure(mode, keep_text=False):
        """K timed steps of one partition mode, bracketed by barrier + synchronize; the MAX over ranks is the job's time."""
        import numpy as np
        mine, my_guides, passes_per_step, bases_per_step_total = partition_mode(mode)
        contig_mode = world > 1 and mode in ("contigs", "windows")
        params_rank = params
        resident = None
        if world > 1 and mode == "windows":
            step_w = 1000 - (len(GUIDE0) + params_kw["max_guide_diffs"] + params_kw["max_gaps_between_guide_and_pam"] - 1)
            first_w, n_w = shard.window_partition(lengths, world, step_w)[rank]
            # every rank holds the contigs its range touches and nothing else (a bin's halo never leaves a contig); rank 0 keeps
            # the whole genome: it checks the gathered file and the sharded batch against single-process searches after the timed region
            if rank != 0 and os.environ.get("CALITAS_BENCH_RESIDENT", "1") != "0":
                resident = shard.resident_contigs(lengths, step_w, first_w, n_w)
        ctx, names, seqs = make_context(mine, resident)
        ranks_block = None
        if world > 1:
            # what every rank holds and runs with: the line says it (worker threads sized from the CPU list and the cgroup quota / ranks)
            import torch.distributed as dist
            info = dict(rank_info, worker_threads=int(os.environ.get("CALITAS_THREADS", "0") or 0), packed_bytes=ctx.reference_info()["packed_bytes"])
            got = [None] * world
            dist.all_gather_object(got, info, group=gloo)
            ranks_block = {"worker_threads": [g.get("worker_threads") for g in got], "packed_bytes": [g.get("packed_bytes") for g in got],
                           "resident_contigs": [g.get("resident_contigs", len(names)) for g in got]}
            rank_info.clear()
        if world > 1 and mode == "windows":
            params_rank = C.make_params(first_window=first_w, n_windows=n_w, **params_kw)
            loads = [shard.range_bases(lengths, step_w, 1000, f, n) for f, n in shard.window_partition(lengths, world, step_w)]
            log("window partition: bases per rank max / mean = %.4f" % (max(loads) / (sum(loads) / world)))
        G = [C.Guide(g) for g in my_guides]
        ids = ["bench%d" % i for i in range(len(G))]
        phase = {"search_hits": 0.0, "gather": 0.0, "search": 0.0, "hits": 0.0, "free": 0.0, "free_text": 0.0}
        # Contig partition, host-side gather without a second copy: one file in shared memory holds a size table and a fixed slot per
        # rank; every rank maps it, page-locks its slot and has the library deliver its piece of hits.txt straight into the slot
        # (calitas_search_hits_into); hits.txt = rank 0's slot followed by the other slots without their header line.
        shm = None
        partition_check = {}
        if contig_mode and len(G) == 1 and args.config == 3:
            import mmap
            import ctypes
            slot_bytes = max(8 << 20, int(160e6 * args.scale / world) * 2)
            slot_bytes = (slot_bytes + 4095) & ~4095
            shm_path = "/dev/shm/calitas_bench_hits_%s.bin" % os.environ.get("MASTER_PORT", "0")
            if rank == 0:
                with open(shm_path, "wb") as f:
                    f.truncate(4096 + world * slot_bytes)
            import torch.distributed as dist
            dist.barrier(group=gloo)
            fd = os.open(shm_path, os.O_RDWR)
            mm = mmap.mmap(fd, 4096 + world * slot_bytes)
            os.close(fd)
            table = np.frombuffer(mm, dtype=np.uint64, count=world * 2)      # per rank: bytes, rows of its last step
            base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
            slot_addr = base + 4096 + rank * slot_bytes
            ctx.pin_host(slot_addr, slot_bytes)
            shm = dict(mm=mm, table=table, slot_addr=slot_addr, slot_bytes=slot_bytes, path=shm_path)
        vcf_path, n_variants = None, 0
        if args.config == 5:
            vcf_path = "/dev/shm/calitas_bench_c5_%d_%d.vcf" % (os.getpid(), rank)
            t_v = time.perf_counter()
            n_variants = synthetic_vcf(vcf_path, names, seqs)
            log("synthetic VCF: %d variants in %.1f s" % (n_variants, time.perf_counter() - t_v))

        def total_rows(rows):
            if not contig_mode:
                return rows
            import torch.distributed as dist
            t = torch.tensor([rows], dtype=torch.int64)
            dist.all_reduce(t, group=gloo)
            return int(t.item())

        # config 5: the text (21.8 GB at full size) goes to a page-locked buffer of this process, as the headline's does
        # (calitas_search_variants_into; --text-block library: a block of the library's per call, as before round 5)
        c5_dst = {"addr": None, "cap": 0}
        if args.config == 5 and args.text_block == "caller":
            t_b = time.perf_counter()
            n_first, _, _ = ctx.search_variants_raw(G[0], "bench", params, vcf_path, "bench", "bench")   # (how large the text is)
            C._lib.lib.calitas_reap_wait()
            cap = int(n_first * 1.02) + (64 << 20)
            c5_dst["addr"], c5_dst["cap"] = C.Context.alloc_host(cap), cap   # (hipHostMalloc: the copy engines write into it directly)
            log("config 5: %d bytes of text; a page-locked destination of %d bytes in %.1f s" % (n_first, cap, time.perf_counter() - t_b))

        def step():
            tp0 = time.perf_counter()
            if args.config == 5:
                if c5_dst["addr"] is not None:
                    text, rows, nwin = ctx.search_variants_into(G[0], "bench", params, vcf_path, c5_dst["addr"], c5_dst["cap"], "bench", "bench")
                else:
                    text, rows, nwin = ctx.search_variants_raw(G[0], "bench", params, vcf_path, "bench", "bench")
                    phase["free_text"] += ctx.last_free_ms / 1e3   # (inside search_hits: calitas_free of the text block)
                tp1 = time.perf_counter()
                tm = ctx.timing()
                tm["hits_bytes"] = text
                tm["variant_windows"] = nwin
                phase["search_hits"] += tp1 - tp0
                return tm, tm["accepted_alignments"], total_rows(rows)
            if len(G) > 1 and not (args.no_hits or args.two_stage):
                # a batch of guides, pipelined through the device stages (calitas_search_hits_batch)
                res = ctx.search_hits_batch(G, ids, params, "bench", "bench", decode=False)
                tp1 = time.perf_counter()
                tm = ctx.timing()
                phase["search_hits"] += tp1 - tp0
                return tm, tm["accepted_alignments"], total_rows(sum(r for _, r in res))
            if shm is not None and not (args.no_hits or args.two_stage):
                nbytes, rows = ctx.search_hits_into(G[0], "bench", params_rank, shm["slot_addr"], shm["slot_bytes"], "bench", "bench")
                tp1 = time.perf_counter()
                tm = ctx.timing()
                shm["table"][2 * rank] = nbytes; shm["table"][2 * rank + 1] = rows      # the gather: the text is in place already
                phase["search_hits"] += tp1 - tp0; phase["gather"] += time.perf_counter() - tp1
                return tm, tm["accepted_alignments"], rows
            if not (args.no_hits or args.two_stage):
                # calitas_search_hits: kernels through to the finished hits.txt text, one copy-back
                text, rows = ctx.search_hits(G[0], "bench", params, "bench", "bench", decode=False)
                tp1 = time.perf_counter()
                tm = ctx.timing()
                phase["search_hits"] += tp1 - tp0
                return tm, tm["accepted_alignments"], rows
            out, n = ctx.search_raw(G, params)
            tp1 = time.perf_counter()
            try:
                tm = ctx.timing()
                rows = 0
                if not args.no_hits:
                    text, rows = ctx.hits_tsv_raw(G[0], "bench", params, out, n, "bench", "bench", decode=False)
            finally:
                tp2 = time.perf_counter()
                C._lib.lib.calitas_free(out)
                tp3 = time.perf_counter()
                phase["search"] += tp1 - tp0; phase["hits"] += tp2 - tp1; phase["free"] += tp3 - tp2
            return tm, n, total_rows(rows)

        # Prime the context before the contract's warmup: the first calls create the lanes, grow the device buffers and the pinned
        # text buffer (three calls); none of it is per-step work.
        for _ in range(3 if args.config == 3 else 1):
            step()
        # ... and the device is at its sustained clocks: the first process on an idle box measured 2.52-2.54 ms per pass after six
        # calls, the same command right after it 2.35.  Untimed, like the contract's warm-up steps that follow.
        # (a fixed number of calls when there are several ranks: a step may contain a collective, so all ranks make the same calls)
        if world > 1:
            for _ in range(int(args.prime_seconds * 400)):
                step()
        else:
            t_prime = time.perf_counter() + args.prime_seconds
            while time.perf_counter() < t_prime:
                step()
        for _ in range(args.warmup):
            step()
        for k in phase:
            phase[k] = 0.0
        sync()
        t0 = time.perf_counter()
        acc = {"scan": 0.0, "align": 0.0, "post": 0.0, "gpu": 0.0, "hitsk": 0.0, "copy": 0.0}
        last = None
        for _ in range(args.steps):
            tm, n_alns, rows = step()
            acc["scan"] += tm["scan_kernel_ms"]; acc["align"] += tm["align_kernel_ms"]; acc["post"] += tm["host_post_ms"]
            acc["gpu"] += tm["gpu_total_ms"]; acc["hitsk"] += tm["hits_kernel_ms"]; acc["copy"] += tm["hits_copy_ms"]
            last = (tm, n_alns, rows)
        # memory the calls handed back on the library's own thread (a 22 GB text per config-5 step, the variant half's tables) is part of
        # the step: the timed region ends when that thread has nothing left to do
        if hasattr(C._lib.lib, "calitas_reap_wait"):
            C._lib.lib.calitas_reap_wait()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        text = None
        if keep_text and rank == 0 and args.config != 5:
            if len(G) > 1:
                text = [t for t, _ in ctx.search_hits_batch(G[:2], ids[:2], params, "bench", "bench")]
            else:
                text = [ctx.search_hits(G[0], "bench", params, "bench", "bench")[0]]
        tiles = ctx.tile_census()
        if c5_dst["addr"] is not None:
            C.Context.free_host(c5_dst["addr"])
        if shm is not None:
            import torch.distributed as dist
            dist.barrier(group=gloo)
            total = 0
            if rank == 0:   # the gathered file of the last step: one header, all rows, contigs in dictionary order
                mm, tb = shm["mm"], shm["table"]
                n_lines, n_bytes = 0, 0
                for r in range(world):
                    lo = 4096 + r * shm["slot_bytes"]
                    piece = mm[lo:lo + int(tb[2 * r])]
                    if r:
                        piece = piece[piece.index(b"\n") + 1:]
                    n_lines += piece.count(b"\n"); n_bytes += len(piece)
                total = int(sum(int(tb[2 * r + 1]) for r in range(world)))
                assert n_lines == total + 1, (n_lines, total)
                log("%s partition: hits.txt gathered in shared memory has %d rows (%d bytes) from %d ranks" % (mode, total, n_bytes, world))
                if mode == "windows":
                    # rank 0 holds the whole genome: the gathered file against ONE process's search, byte for byte (outside the timed region)
                    import zlib
                    gathered = b"".join((mm[4096 + r * shm["slot_bytes"]: 4096 + r * shm["slot_bytes"] + int(tb[2 * r])] if r == 0 else
                                         mm[4096 + r * shm["slot_bytes"]: 4096 + r * shm["slot_bytes"] + int(tb[2 * r])].split(b"\n", 1)[1]) for r in range(world))
                    single, n_single = ctx.search_hits(G[0], "bench", params, "bench", "bench", decode="bytes")
                    partition_check.update(rows=n_single, crc_single=zlib.crc32(single), crc_gathered=zlib.crc32(gathered), identical=bool(single == gathered))
                    log("window partition: gathered text identical to a single-process search: %s (%d rows)" % (single == gathered, n_single))
            ctx.unpin_host(shm["slot_addr"])
            dist.barrier(group=gloo)
            del shm["table"]
            if rank == 0:
                os.unlink(shm["path"])
                last = (last[0], last[1], total)
        batch_block = None
        want_batch = args.batch_sharded if args.batch_sharded is not None else world > 1
        if want_batch and args.config == 3 and mode in ("windows", "none") and not (args.no_hits or args.two_stage):
            batch_block = measure_batch_sharded(ctx, params_rank, names, seqs)
        ctx.close()
        if vcf_path:
            os.unlink(vcf_path)
        return dict(dt=dt, acc=acc, batch_block=batch_block, last=last, phase=phase, my_guides=my_guides, passes_per_step=passes_per_step,
                    bases_per_step_total=bases_per_step_total, names=names, seqs=seqs, text=text, tiles=tiles, n_variants=n_variants,
                    mine=mine, partition_check=partition_check, ranks=ranks_block)

    headline_mode = args.shard if world > 1 else "none"
    if headline_mode == "windows" and args.config != 3:
        headline_mode = "contigs"             # the guide batch and the variant branch are partitioned by contigs
    m = measure(headline_mode, keep_text=(world == 1))
    second = None
    if world > 1 and args.secondary and not args.no_secondary and args.config == 3:
        other = "guides" if headline_mode in ("contigs", "windows") else "contigs"
        s = measure(other)
        second = {"partition": other, "scaling": "weak" if other == "guides" else "strong",
                  "value": 2.0 * s["bases_per_step_total"] * args.steps / s["dt"], "unit": "candidates/s", "ms_per_step": s["dt"] / args.steps * 1e3,
                  "guide_passes_per_step": s["passes_per_step"],
                  "note": ("every rank holds the whole genome and runs its own guide pass (the shape of a guide batch spread over the GPUs)"
                           if other == "guides" else "consecutive contig ranges of one pass, one range per rank")}

    if rank == 0:
        K = args.steps
        dt, acc = m["dt"], m["acc"]
        tm, n_alns, rows = m["last"]
        value = 2.0 * m["bases_per_step_total"] * K / dt
        # a pass is scanned in `launches` launches (one per contig range / guide); per-launch figures are averages over them
        n_guides_rank = len(m["my_guides"])
        launches = max(1, int(tm.get("scan_launches") or tm.get("lanes", 1))) if n_guides_rank == 1 else n_guides_rank
        if tm.get("contig_passes", 0):
            launches = int(tm["contig_passes"])
        scan_avg_ms = acc["scan"] / K / launches
        packed_rank = tm["packed_bytes"] / max(1, n_guides_rank)            # algorithmic bytes of ONE pass over this rank's slice
        bytes_per_launch = packed_rank / (launches if n_guides_rank == 1 else 1)
        achieved = bytes_per_launch / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0
        tiles = m["tiles"]
        result = {
            "metric": "off-target candidates/sec (hg38 full scan), 20nt guide+NRG PAM",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "strong" if (world > 1 and headline_mode in ("contigs", "windows")) else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": args.config,
                       "guide": m["my_guides"][0], "guides_per_step_per_rank": n_guides_rank, "guide_passes_per_step": m["passes_per_step"],
                       "partition": headline_mode, "genome_scale": args.scale,
                       "step_includes": "scan + align kernels, per-window filter" + ("" if args.no_hits else ", removeOverlaps, sort, all hits.txt rows")
                                        + (", copy-back of alignments, host rows" if (args.two_stage or args.no_hits or args.config == 5)
                                           else " (all on the device), copy-back of the text")},
            "bases_per_s": m["bases_per_step_total"] * K / dt,
            "dead_tile_fraction": tiles["dead"] / max(1, tiles["tiles"]),
            "hits_per_pass": rows, "accepted_alignments_per_pass": n_alns, "raw_alignments_per_pass": tm["raw_alignments"],
            "scan_records_per_pass": tm["scan_records"],
            "hits_bytes_per_pass": tm["hits_bytes"],
            "host_phase_ms": {k: v / K * 1e3 for k, v in m["phase"].items() if v > 0},
            "kernel_ms": {"scan": acc["scan"] / K, "align": acc["align"] / K, "search_gpu_total_sum_over_lanes": acc["gpu"] / K, "hits_kernels": acc["hitsk"] / K,
                          "text_copy": acc["copy"] / K, "host_convert": acc["post"] / K},
            "roofline": {"bound": "hbm", "kernel": "scan_rows_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": scan_avg_ms, "launches_per_step": launches,
                         "note": "integer-VALU bound by construction (row-wise bit-vector edit-distance filter: 10 vector instructions per "
                                 "32 DP cells, 20 rows, 2 strands); roofline.valu is the roof that binds, see DESIGN.md 4.1"},
        }
        if args.config == 5:
            result["config"]["variants"] = m["n_variants"]
            result["config"]["variant_windows_per_pass"] = tm.get("variant_windows", 0)
            result["config"]["text_destination"] = ("a page-locked block (calitas_alloc_host) of the caller's, reused from step to step (calitas_search_variants_into)"
                                                    if args.text_block == "caller" else "a block of the library's per step (calitas_search_variants), freed inside the step")
        if args.config == 4:
            result["roofline"]["g_equivalent"] = {"guides_per_step": n_guides_rank, "achieved": achieved,
                                                  "note": "every guide of the batch is a scan launch of its own over the whole slice: bytes are counted once per launch, "
                                                          "nothing is shared between guides"}
        # instruction count and HBM traffic of the dominant kernel for this exact workload, from the committed counter passes
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_summary.json")))["scan_rows_kernel"]
            if world == 1 and args.config in (3, 4) and args.scale == 1.0:
                scale = 1.0 if n_guides_rank > 1 else 1.0 / launches
                result["roofline"]["traffic"] = pm["traffic_bytes_per_pass"] * scale
                result["roofline"]["traffic_source"] = "profiles/r05_pmc_summary.json"
                result["roofline"]["traffic_measured_in_run"] = False
                insts = pm["SQ_INSTS_VALU"]                                 # wave-instructions of one pass (one guide)
                ach = insts * max(1, n_guides_rank) / (acc["scan"] / K * 1e6)   # per nanosecond
                result["roofline"]["valu"] = {
                    "bound": "valu-issue", "achieved": ach, "unit": "G wave-inst/s", "wave_insts_per_pass": insts,
                    "lane_ops_per_base": pm["valu_lane_ops_per_base"], "counted_in_run": False, "source": "profiles/r05_pmc_summary.json",
                    "peak": VALU_PEAK_GUIDE, "frac": ach / VALU_PEAK_GUIDE,
                    "peak_note": "256 CU x 4 SIMD-32, one wave64 instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)",
                    "peak_measured": 1024 / pm["valu_full_rate_ns_per_wave_inst_per_simd"],
                    "frac_of_measured": ach / (1024 / pm["valu_full_rate_ns_per_wave_inst_per_simd"]),
                    "peak_measured_note": "v_and / v_add streams on all 1024 SIMDs, profiles/r01_valu_rates.txt"}
                # What the kernel's OWN instruction mix can issue at: not every integer instruction runs at the v_and rate on gfx950.  The
                # census comes from the compiler's output for scan_rows_kernel<8, 1> and the per-instruction issue times from the
                # micro-benchmarks (tools/scan_census.py -> profiles/r04_scan_census.json; rates: profiles/valu_rates.json): per wave, a strand
                # of a tile = L rows of the row loop + the rest of the guide x strand (the bottom-row test) + its share of what runs once.
                cen = json.load(open(os.path.join(ROOT, "profiles", "r04_scan_census.json")))
                L_rows = len(GUIDE0) - 3 if args.config != 5 else 20
                row_ns = sum(cen["ns_per_word_row"]) / len(cen["ns_per_word_row"])
                strand_ns = (sum(L_rows * r + g for r, g in zip(cen["row_iteration_ns"], cen["guide_strand_rest_ns"])) + cen["once_per_wave_ns"]) / 2
                bases_per_ns = 1024 * (64 * 8 * 32) / (2 * strand_ns)
                live = 1.0 - tiles["dead"] / max(1, tiles["tiles"])
                mix_ms = sum(lengths) * live / bases_per_ns * 1e-6
                result["roofline"]["valu"]["mix_limit"] = {
                    "ns_per_word_row": round(row_ns, 2), "scan_ms_at_limit": round(mix_ms, 4),
                    "frac": round(mix_ms / (acc["scan"] / K / max(1, n_guides_rank)), 4),
                    "valu_per_word_row": round(sum(cen["valu_per_row_iteration"]) / len(cen["valu_per_row_iteration"]) / cen["words_per_lane"], 2),
                    "census": "profiles/r04_scan_census.json (tools/scan_census.py: the compiler's output for " + cen["kernel"] + ")",
                    "note": "issue time of the kernel's instruction mix at the per-instruction rates of profiles/valu_rates.json; frac = that "
                            "time / the measured scan time per pass (sum over the launches of a call, which share the chip with the tails)"}
        except Exception:
            pass
        try:
            # what stands behind the scan (expand, align, trace, the filter / hits / row kernels): the chip time of a guide's tail as the
            # events give it, and how much of the chip's issue rate its instructions (counted in a committed counter pass) would need
            tc = json.load(open(os.path.join(ROOT, "profiles", "r05_tail_census.json")))
            tail_ms = (acc["align"] + acc["hitsk"]) / K / max(1, n_guides_rank)
            winst = tc["total"]["valu_wave_inst_per_pass"]
            result["roofline"]["tail"] = {
                "kernel_ms_per_guide": round(tail_ms, 4), "valu_wave_inst_per_guide": winst,
                "issue_frac": round(winst / (tail_ms * 1e-3 * 1228.8e9), 4) if tail_ms > 0 else None,
                "counted_in_run": False, "source": "profiles/r05_tail_census.json",
                "note": "kernel_ms = sum over the call's lanes of the kernels behind the scan (events / device stamps); the instruction count is guide #0's "
                        "per hg38-sized pass on the per-bin tail (the random 20-mers of config 4 have ~3x its records); issue_frac = instructions / (kernel_ms x "
                        "1228.8 G wave-inst/s): how far the tail is from being bound by issue -- it is bound by waiting"}
        except Exception:
            pass
        if m.get("batch_block"):
            result["batch_sharded"] = m["batch_block"]          # BASELINE config 4's shape on this partition (labelled; `value` stays the one-pass headline)
            if not m["batch_block"]["check"]["identical"]:
                print(json.dumps(result), flush=True)
                raise SystemExit("bench.py: a rank's text of the sharded batch differs from the single-process search")
        if m.get("ranks"):
            result["ranks"] = m["ranks"]                         # per rank: worker threads, packed reference bytes held, contigs resident
        if m.get("partition_check"):
            result["partition_check"] = m["partition_check"]     # the gathered hits.txt of the N ranks against one process's search
            if not m["partition_check"].get("identical", True):
                print(json.dumps(result), flush=True)
                raise SystemExit("bench.py: the gathered text of the window partition differs from a single-process search")
        if second is not None:
            result["guide_sharded" if second["partition"] == "guides" else "contig_sharded"] = second
        mb = args.cpu_sample_mb
        if mb < 0:
            # the oracle runs ~2.7 Mb/s per core and guide pass on the GPU box: ~15 s of CPU work
            mb = 40.0 * min(host_cores(), 16) / (2 if args.config == 4 else 4 if args.config == 5 else 1)
        if mb > 0 and world == 1 and args.config == 5:
            try:
                report, parity = cpu_baseline_c5(m["names"], m["seqs"], m["my_guides"][0], params_kw, int(mb * 1e6))
                result["cpu_baseline"] = report
                result["parity_sample"] = parity
                if not parity["identical"]:
                    print(json.dumps(result), flush=True)
                    raise SystemExit("bench.py: GPU rows of the variant search differ from the oracle's on the parity sample")
            except SystemExit:
                raise
            except Exception as e:
                result["cpu_baseline"] = {"error": str(e)}
        elif mb > 0 and world == 1:
            try:
                cg = m["my_guides"][:2] if args.config == 4 else m["my_guides"][:1]
                report, whole, oracle_rows = cpu_baseline(m["names"], m["seqs"], cg, params_kw, int(mb * 1e6),
                                                          ["bench%d" % i for i in range(len(cg))] if args.config == 4 else None)
                result["cpu_baseline"] = report
                if m["text"] is not None and whole:
                    # the bench checks its own output: the rows of the sample's whole contigs, every column but the run-dependent two
                    skip = {"aligner_version", "time_stamp", "genome_build"}
                    n_cmp, same = 0, True
                    for gi, text in enumerate(m["text"][:len(oracle_rows)]):
                        got = [{k: v for k, v in r.items() if k not in skip} for r in C.read_hits(text) if r["chromosome"] in whole]
                        want = [{k: v for k, v in r.items() if k not in skip} for r in oracle_rows[gi] if r["chromosome"] in whole]
                        n_cmp += len(want)
                        same = same and got == want
                    result["parity_sample"] = {"contigs": whole, "guides": len(oracle_rows), "rows": n_cmp, "identical": bool(same)}
                    if not same:
                        print(json.dumps(result), flush=True)
                        raise SystemExit("bench.py: GPU rows differ from the oracle's on %s" % ",".join(whole))
            except SystemExit:
                raise
            except Exception as e:  # the baseline is reporting only; never fail the bench line because of it
                result["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
