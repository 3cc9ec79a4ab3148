#!/usr/bin/env python3
"""bench.py -- SearchReference full-scan throughput on MI355X (BASELINE.json metric).

Workload (N=1): BASELINE config 3 -- one 20-nt guide + NRG PAM against a synthetic hg38-sized genome (25 contigs,
3 088 286 401 bp, N runs, soft-masking, tandem repeats, planted sites), max-guide-diffs 5, max-pam-mismatches 1,
max-gaps-between-guide-and-pam 2.  A "step" is one complete SearchReference pass for one guide over the resident
reference: scan kernel + aligner kernel + copy-back + per-window filter + removeOverlaps/sort/hit rows (everything
except writing hits.txt to disk).  The packed reference is resident in HBM before the timed region.

N>1: one process per GPU (torch.distributed, RCCL for the barrier).  Default partition = guides (each rank holds the
genome and runs one guide pass per step -- guide #0 on every rank, so the work per GPU is that of the N=1 line;
--distinct-guides draws rank r's guide from the 96-guide set of BASELINE config 4) -> weak scaling, no data-path
collective.  --shard contigs partitions the contigs of ONE guide's pass instead (strong scaling, host-side gather).

value = candidate loci examined per second = 2 strands x reference bases x guide-passes / wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GUIDE0 = "CTTGCCCCACAGGGCAGTAAnrg"   # README.md:74 of the reference
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


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


def host_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(names, seqs, params_kw, budget_bases):
    """Times the CPU oracle (the restatement of the reference algorithm, oracle/) on a bounded sample of the same
    genome with one worker per host core -- the reference's own threading model (SearchReference.scala:459)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = min(host_cores(), 16)   # the GPU box gives one GPU a 16-core CPU share
    # sample: whole contigs of the same genome, in order, until the budget is reached (the last one truncated)
    s_names, s_seqs, total = [], [], 0
    for n, s in zip(names, seqs):
        if total >= budget_bases:
            break
        take = min(len(s), budget_bases - total)
        s_names.append(n); s_seqs.append(bytes(s[:take])); total += take
    t0 = time.perf_counter()
    _, rows, nwin = O.search_memory(s_names, s_seqs, GUIDE0, "cpu", d=params_kw["max_guide_diffs"],
                                    p=params_kw["max_pam_mismatches"], g=params_kw["max_gaps_between_guide_and_pam"], threads=cores)
    dt = time.perf_counter() - t0
    return {"value": 2 * total / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
            "sample": "%d bp (%s%s; %d windows, %d hits) in %.1f s; oracle/ C++ restatement of the reference algorithm, %d threads"
                      % (total, ",".join(s_names[:3]), "..." if len(s_names) > 3 else "", nwin, len(rows), dt, cores),
            "bases_per_s": total / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="genome size relative to hg38 (1.0 = 3.09 Gb)")
    ap.add_argument("--shard", choices=["guides", "contigs"], default="guides")
    ap.add_argument("--cpu-sample-mb", type=float, default=-1, help="CPU baseline sample in Mb (<0: auto, 0: skip)")
    ap.add_argument("--no-hits", action="store_true", help="time the search only (no removeOverlaps / row building)")
    ap.add_argument("--guides-per-step", type=int, default=1,
                    help="guides each rank runs per step through calitas_search_hits_batch (BASELINE config 4 shape: 96 guides / 8 GPUs = 12); "
                         "the default 1 is the BASELINE metric's single-guide pass")
    ap.add_argument("--same-guide", action="store_true", help="with --guides-per-step: every guide of the batch is guide #0 (isolates the pipelining gain)")
    ap.add_argument("--distinct-guides", action="store_true", help="N>1: rank r runs guide #r of the 96-guide set instead of guide #0 on every rank")
    ap.add_argument("--two-stage", action="store_true", help="calitas_search + calitas_hits_tsv (host rows) instead of the fused calitas_search_hits")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and gloo replaces RCCL (not a measurement)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    params_kw = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    params = C.make_params(**params_kw)
    all_guides = [GUIDE0] + synth.random_guides(0xC4, 95)
    spec = synth.hg38_like_spec(args.scale)
    lengths = [l for _, l in spec]

    if world > 1 and args.shard == "contigs":
        mine = shard.contiguous_partition(lengths, world)[rank]   # consecutive ranges: the gather is a concatenation
        my_guides = [GUIDE0]
        guide_passes_per_step = 1          # the ranks share ONE guide pass
        bases_per_step_total = sum(lengths)
    else:
        mine = None
        gps = max(1, args.guides_per_step)
        # weak scaling keeps the work per GPU fixed: every rank runs the pass of the N=1 line (guide #0).  The 96-guide set of
        # BASELINE config 4 (its random 20-mers yield ~3x the rows of guide #0 on this genome, so a rank's step is copy-back
        # bound and takes longer) is drawn with --distinct-guides, or by a batch (--guides-per-step > 1) without --same-guide.
        distinct = args.distinct_guides or (gps > 1 and not args.same_guide)
        my_guides = [all_guides[(rank * gps + i) % len(all_guides)] if distinct else GUIDE0 for i in range(gps)]
        guide_passes_per_step = world * gps   # every rank runs its own guide(s) over the whole genome
        bases_per_step_total = sum(lengths) * world * gps

    t_gen = time.perf_counter()
    names, seqs = build_genome(args.scale, device, contig_indices=mine, guides=[GUIDE0], log=None)
    log("genome: %d contigs, %d bp on this rank, generated in %.1f s" % (len(names), sum(len(s) for s in seqs), time.perf_counter() - t_gen))
    ctx = C.Context(local_rank)
    t_set = time.perf_counter()
    ctx.set_reference(names, seqs, genome_build="synthetic-hg38-sized")
    info = ctx.reference_info()
    log("set_reference: %.2f s (pack + upload), %d packed bytes" % (time.perf_counter() - t_set, info["packed_bytes"]))

    G = [C.Guide(g) for g in my_guides]

    phase = {"search_hits": 0.0, "gather": 0.0, "search": 0.0, "hits": 0.0, "free": 0.0}

    contig_mode = world > 1 and args.shard == "contigs"

    shm_path = "/dev/shm/calitas_bench_hits_%s.txt" % os.environ.get("MASTER_PORT", "0")
    shm_fd = os.open(shm_path, os.O_RDWR | os.O_CREAT, 0o600) if contig_mode else -1

    def place_rows(view, rows):
        """Contig partition, no copies: this rank's piece of the job's hits.txt goes from the library's buffer straight into the
        shared file at its offset (rank 0 keeps the header line); only the sizes travel between the ranks."""
        import torch.distributed as dist
        nl = 0
        if rank:
            while view[nl] != 10:
                nl += 1
            nl += 1
        sizes = torch.zeros(world, 2, dtype=torch.int64)
        dist.all_gather_into_tensor(sizes.view(-1), torch.tensor([len(view) - nl, rows], dtype=torch.int64), group=gloo)
        os.pwrite(shm_fd, view[nl:], int(sizes[:rank, 0].sum()))
        return int(sizes[:, 1].sum())

    def gather_rows(text, rows):
        """Contig partition: every rank owns a consecutive contig range, so its rows are a consecutive piece of the job's hits.txt.
        The ranks share one node: each writes its piece at its offset into one file in shared memory; only the sizes travel."""
        if not contig_mode:
            return rows
        import torch.distributed as dist
        nl = text.index(b"\n") + 1
        body = text[nl:] if rank else text                    # rank 0 keeps the header line
        sizes = torch.zeros(world, 2, dtype=torch.int64)
        mine_t = torch.tensor([len(body), rows], dtype=torch.int64)
        dist.all_gather_into_tensor(sizes.view(-1), mine_t, group=gloo)
        os.pwrite(shm_fd, body, int(sizes[:rank, 0].sum()))
        return int(sizes[:, 1].sum())

    def step():
        tp0 = time.perf_counter()
        if len(G) > 1 and not (args.no_hits or args.two_stage):
            # a batch of guides, pipelined through the device stages (calitas_search_hits_batch)
            res = ctx.search_hits_batch(G, ["bench%d" % i for i in range(len(G))], params, "bench", "bench", decode=False)
            tp1 = time.perf_counter()
            tm = ctx.timing()
            phase["search_hits"] += tp1 - tp0
            return tm, tm["accepted_alignments"], sum(r for _, r in res)
        if contig_mode and not (args.no_hits or args.two_stage):
            with ctx.search_hits_view(G[0], "bench", params, "bench", "bench") as (view, rows):
                tp1 = time.perf_counter()
                tm = ctx.timing()
                rows = place_rows(view, rows)
            phase["search_hits"] += tp1 - tp0; phase["gather"] += time.perf_counter() - tp1
            return tm, tm["accepted_alignments"], rows
        if not (args.no_hits or args.two_stage):
            # calitas_search_hits: kernels through to the finished hits.txt text, one copy-back
            text, rows = ctx.search_hits(G[0], "bench", params, "bench", "bench", decode=False)
            tp1 = time.perf_counter()
            tm = ctx.timing()
            rows = gather_rows(text, rows)
            phase["search_hits"] += tp1 - tp0; phase["gather"] += time.perf_counter() - tp1
            return tm, tm["accepted_alignments"], rows
        out, n = ctx.search_raw(G, params)
        tp1 = time.perf_counter()
        try:
            tm = ctx.timing()
            rows = 0
            if not args.no_hits:
                text, rows = ctx.hits_tsv_raw(G[0], "bench", params, out, n, "bench", "bench", decode=contig_mode)
                if contig_mode:
                    text = text.encode()
                rows = gather_rows(text, rows)
        finally:
            tp2 = time.perf_counter()
            C._lib.lib.calitas_free(out)
            tp3 = time.perf_counter()
            phase["search"] += tp1 - tp0; phase["hits"] += tp2 - tp1; phase["free"] += tp3 - tp2
        return tm, n, rows

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Prime the context before the contract's warmup: the first calls create the lanes, grow the device buffers and the pinned
    # text buffer, and the runtime sets up its copy path under load (three calls); none of it is per-step work.
    for _ in range(3):
        step()
    for _ in range(args.warmup):
        step()
    for k in phase:
        phase[k] = 0.0
    sync()
    t0 = time.perf_counter()
    scan_ms = align_ms = post_ms = gpu_ms = hitsk_ms = copy_ms = 0.0
    last = None
    for _ in range(args.steps):
        tm, n_alns, rows = step()
        scan_ms += tm["scan_kernel_ms"]; align_ms += tm["align_kernel_ms"]; post_ms += tm["host_post_ms"]; gpu_ms += tm["gpu_total_ms"]
        hitsk_ms += tm["hits_kernel_ms"]; copy_ms += tm["hits_copy_ms"]
        last = (tm, n_alns, rows)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        tm, n_alns, rows = last
        K = args.steps
        bases_rank = tm["bases_scanned"]
        value = 2.0 * bases_per_step_total * K / dt
        # a pass is scanned in `lanes` launches (one per contig range); per-launch figures are averages over them
        lanes = max(1, int(tm.get("lanes", 1)))
        if len(G) > 1:
            lanes = len(G)      # a batch scans the whole reference once per guide: one launch each
        scan_avg_ms = scan_ms / K / lanes
        bytes_per_launch = tm["packed_bytes"] / lanes
        achieved = bytes_per_launch / (scan_avg_ms * 1e-3) / 1e9    # GB/s, algorithmic bytes of one launch / its duration
        result = {
            "metric": "off-target candidates/sec (hg38 full scan), 20nt guide+NRG PAM",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "strong" if (world > 1 and args.shard == "contigs") else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "SearchReference: 20 nt guide + NRG PAM vs synthetic hg38-sized genome (25 contigs, %d bp), "
                                   "max-guide-diffs=5 max-pam-mismatches=1 max-gaps-between-guide-and-pam=2" % sum(lengths),
                       "guide": my_guides[0], "guide_passes_per_step": guide_passes_per_step, "partition": args.shard if world > 1 else "none",
                       "genome_scale": args.scale,
                       "step_includes": "scan + align kernels, per-window filter" + ("" if args.no_hits else ", removeOverlaps, sort, all hits.txt rows")
                                        + (", copy-back of alignments, host rows" if (args.two_stage or args.no_hits) else " (all on the device), copy-back of the text")},
            "bases_per_s": bases_per_step_total * K / dt,
            "hits_per_pass": rows, "accepted_alignments_per_pass": n_alns, "raw_alignments_per_pass": tm["raw_alignments"],
            "scan_records_per_pass": tm["scan_records"],
            "hits_bytes_per_pass": tm["hits_bytes"],
            "host_phase_ms": {k: v / K * 1e3 for k, v in phase.items() if v > 0},
            "kernel_ms": {"scan": scan_ms / K, "align": align_ms / K, "search_gpu_total": gpu_ms / K, "hits_kernels": hitsk_ms / K,
                          "text_copy": copy_ms / K, "host_convert": post_ms / K},
            "roofline": {"bound": "hbm", "kernel": "scan_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": scan_avg_ms, "launches_per_step": lanes,
                         "note": "integer-VALU bound by construction (bit-vector edit-distance filter, ~31 int lane-ops per base for "
                                 "two strands); see DESIGN.md 4.1 for the VALU-side roofline"},
        }
        # measured HBM traffic of the dominant kernel for this exact workload, from the committed PMC passes
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))["scan_kernel"]
            if pm["workload_packed_bytes"] * len(G) == tm["packed_bytes"] and world == 1:
                result["roofline"]["traffic"] = pm["traffic_bytes_per_launch"] / (1 if len(G) > 1 else lanes)   # measured on the one-launch pass
                result["roofline"]["traffic_source"] = "profiles/r01_pmc_summary.json"
            if "SQ_INSTS_VALU" in pm and pm["workload_packed_bytes"] == tm["packed_bytes"] and world == 1 and len(G) == 1:
                # the roof that binds this kernel: vector-ALU instruction issue.  Counted instructions per
                # pass (PMC) / measured scan time of this run, against the measured full-rate issue of all 1024 SIMDs.
                peak = 1024 / pm["valu_full_rate_ns_per_wave_inst_per_simd"]            # wave-instructions per ns
                ach = pm["SQ_INSTS_VALU"] / (scan_ms / K * 1e6)
                result["roofline"]["valu"] = {"bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "G wave-inst/s",
                                              "frac": ach / peak, "wave_insts_per_pass": pm["SQ_INSTS_VALU"],
                                              "lane_ops_per_base": pm["valu_lane_ops_per_base"]}
        except Exception:
            pass
        mb = args.cpu_sample_mb
        if mb < 0:
            mb = 40.0 * min(host_cores(), 16)   # the oracle runs ~2.7 Mb/s per core on the GPU box: ~15 s of CPU work
        if mb > 0 and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(names, seqs, params_kw, int(mb * 1e6))
            except Exception as e:  # the baseline is reporting only; never fail the bench line because of it
                result["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        if contig_mode:
            dist.barrier(group=gloo)
            if rank == 0:   # the assembled file of the last step: one header, `rows` rows, contigs in dictionary order
                size = os.fstat(shm_fd).st_size
                whole = os.pread(shm_fd, size, 0)
                n_lines = whole.count(b"\n")
                log("contig partition: assembled hits.txt in shared memory has %d lines (%d bytes)" % (n_lines, size))
                os.unlink(shm_path)
            os.close(shm_fd)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
