"""Host-side time line of calitas_search_hits on the bench genome (CALITAS_TRACE=2): one line of marks per lane thread, in
microseconds since the start of the call.  python tools/trace_marks.py [scale] [calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    import torch
    import bench
    import calitas_amd as C
    names, seqs = bench.build_genome(scale, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="synthetic")
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = C.Guide(bench.GUIDE0)
    for _ in range(3):
        ctx.search_hits(G, "a", params, "v0", "stamp", decode="bytes")
    os.environ["CALITAS_TRACE"] = "2"
    for i in range(calls):
        sys.stderr.write("--- call %d\n" % i)
        sys.stderr.flush()
        ctx.search_hits(G, "a", params, "v0", "stamp", decode="bytes")
    ctx.close()


if __name__ == "__main__":
    main()
