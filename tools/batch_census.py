"""The vector instructions of a guide batch (BASELINE config 4's settings), kernel by kernel, per guide.  Two modes:
  python3 tools/batch_census.py run N        -- N guides through calitas_search_hits_batch, twice (what rocprofv3 --pmc wraps)
  python3 tools/batch_census.py sum CSV N    -- sums of the counter_collection.csv of such a run, per guide (2 N guide passes)
bash tools/pmc_batch.sh does both inside gpurun."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n):
    # (no torch here: its runtime does not survive rocprofv3 --pmc on this image; the genome is tools/scan_profile.py's)
    import numpy as np
    import time
    import calitas_amd as C
    from calitas_amd import synth
    GUIDE0 = "CTTGCCCCACAGGGCAGTAAnrg"
    names, seqs = [], []
    t0 = time.time()
    for ci, (name, length) in enumerate(synth.hg38_like_spec(1.0)):
        rng = np.random.default_rng([0xC3, ci])
        s = synth.random_bases(rng, length)
        if length > 100000:
            s[:10000] = ord("N"); s[-10000:] = ord("N")
            s[length // 2: length // 2 + length // 100] = ord("N")
        names.append(name); seqs.append(s)
    print("genome %d bp in %.1f s" % (sum(len(s) for s in seqs), time.time() - t0), flush=True)
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="x")
    del seqs
    guides = ([GUIDE0] + synth.random_guides(0xC4, 95))[:n]
    G = [C.Guide(g) for g in guides]
    ids = ["g%02d" % i for i in range(len(G))]
    p = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    for _ in range(2):
        res = ctx.search_hits_batch(G, ids, p, "v", "t", decode=False)
        print("rows", sum(r for _, r in res), "bytes", sum(b for b, _ in res), flush=True)
        del res
    ctx.close()


def summarise(path, n):
    import csv
    per = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("calitas::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "rocprim" in k:
            k = "rocprim (all)"
        if k.startswith("at::") or "planes_kernel" in k or "window_table" in k or "dpp_selftest" in k:
            continue
        d = per.setdefault(k[:48], {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            d["launches"] = d.get("launches", 0) + 1
    passes = 2.0 * n
    tot = sum(d.get("SQ_INSTS_VALU", 0.0) for d in per.values())
    print("per guide pass (%d guides x 2 calls): %-30s %12s %12s %10s %8s" % (n, "kernel", "VALU", "SALU", "waves", "launches"))
    for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0.0)):
        print("%-70s %12.4g %12.4g %10.0f %8.2f  %5.1f %%" % (k, d.get("SQ_INSTS_VALU", 0) / passes, d.get("SQ_INSTS_SALU", 0) / passes,
                                                           d.get("SQ_WAVES", 0) / passes, d.get("launches", 0) / passes, 100.0 * d.get("SQ_INSTS_VALU", 0) / tot))
    print("%-70s %12.4g" % ("all", tot / passes))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        summarise(sys.argv[2], int(sys.argv[3]))
