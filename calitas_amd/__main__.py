"""`python -m calitas_amd <Tool> [flags]` -- the reference's four tools (Main.scala; flags as in SearchReference.scala:452-470,
AlignToReference.scala:34-51, PairwiseAlignSequences.scala:25-33, PrepareVcf.scala:32-36) on the MI355X path.
`-t/--threads` is accepted and ignored: the GPU replaces the thread pool."""
import argparse
import sys

from .aligner import Defaults, SearchReference
from .tools import align_to_reference, pairwise_align_sequences
from .variants import prepare_vcf


def _costs(ap):
    ap.add_argument("-m", "--guide-mismatch-net-cost", type=int, default=Defaults.MismatchNetCost)
    ap.add_argument("-M", "--pam-mismatch-net-cost", type=int, default=Defaults.PamMismatchNetCost)
    ap.add_argument("-b", "--genome-gap-net-cost", type=int, default=Defaults.GenomeGapNetCost)
    ap.add_argument("-B", "--guide-gap-net-cost", type=int, default=Defaults.GuideGapNetCost)
    ap.add_argument("-t", "--threads", type=int, default=8)
    ap.add_argument("--device", type=int, default=0, help="HIP device index")


def main(argv=None):
    top = argparse.ArgumentParser(prog="calitas_amd")
    sub = top.add_subparsers(dest="tool", required=True)

    sr = sub.add_parser("SearchReference")
    sr.add_argument("-i", "--guide", required=True)
    sr.add_argument("-I", "--guide-id", required=True)
    sr.add_argument("-x", "--auxiliary-pams", nargs="*", default=[])
    sr.add_argument("-r", "--ref", required=True)
    sr.add_argument("-v", "--variants")
    sr.add_argument("-V", "--max-variants", type=int, default=Defaults.MaxVariantsInCluster)
    sr.add_argument("-o", "--output")
    sr.add_argument("-w", "--window-size", type=int, default=1000)
    sr.add_argument("-d", "--max-guide-diffs", type=int, default=Defaults.MaxGuideDiffs)
    sr.add_argument("-p", "--max-pam-mismatches", type=int, default=Defaults.MaxPamMismatches)
    sr.add_argument("-g", "--max-gaps-between-guide-and-pam", type=int, default=Defaults.MaxGapsBetweenGuideAndPam)
    sr.add_argument("-D", "--max-total-diffs", type=int)
    sr.add_argument("-O", "--max-overlap", type=int, default=Defaults.MaxOverlap)
    sr.add_argument("-c", "--chrom")
    _costs(sr)

    a2r = sub.add_parser("AlignToReference")
    a2r.add_argument("-i", "--input", required=True)
    a2r.add_argument("-r", "--ref", required=True)
    a2r.add_argument("-o", "--output")
    a2r.add_argument("-w", "--window-size", type=int)
    a2r.add_argument("-d", "--max-guide-diffs", type=int)
    a2r.add_argument("-p", "--max-pam-mismatches", type=int)
    a2r.add_argument("-g", "--max-gaps-between-guide-and-pam", type=int, default=Defaults.MaxGapsBetweenGuideAndPam)
    a2r.add_argument("-D", "--max-total-diffs", type=int)
    a2r.add_argument("-O", "--max-overlap", type=int)
    _costs(a2r)

    pas = sub.add_parser("PairwiseAlignSequences")
    pas.add_argument("-i", "--input", required=True)
    pas.add_argument("-o", "--output", default="/dev/stdout")
    pas.add_argument("-g", "--max-gaps-between-guide-and-pam", type=int, default=Defaults.MaxGapsBetweenGuideAndPam)
    pas.add_argument("-O", "--max-overlap", type=int, default=Defaults.MaxOverlap)   # declared and unused by the reference as well
    _costs(pas)

    pv = sub.add_parser("PrepareVcf")
    pv.add_argument("-i", "--input", nargs="+", required=True)
    pv.add_argument("-o", "--output", required=True)
    pv.add_argument("-f", "--min-af", type=float, default=0.01)
    pv.add_argument("-d", "--dict", help="sequence dictionary (.dict, .fai, or a FASTA with its .dict) that overrides the contig lines")
    pv.add_argument("-c", "--add-chr-prefix", type=lambda s: s.lower() in ("1", "true", "yes"), default=True)

    a = top.parse_args(argv)
    if a.tool == "SearchReference":
        SearchReference(guide=a.guide, guide_id=a.guide_id, ref=a.ref, output=a.output, auxiliary_pams=a.auxiliary_pams,
                        window_size=a.window_size, max_guide_diffs=a.max_guide_diffs, max_pam_mismatches=a.max_pam_mismatches,
                        max_gaps_between_guide_and_pam=a.max_gaps_between_guide_and_pam, max_total_diffs=a.max_total_diffs,
                        max_overlap=a.max_overlap, guide_mismatch_net_cost=a.guide_mismatch_net_cost,
                        pam_mismatch_net_cost=a.pam_mismatch_net_cost, genome_gap_net_cost=a.genome_gap_net_cost,
                        guide_gap_net_cost=a.guide_gap_net_cost, chrom=a.chrom, variants=a.variants, max_variants=a.max_variants,
                        device=a.device).execute()
    elif a.tool == "AlignToReference":
        text = align_to_reference(a.input, a.ref, a.output, window_size=a.window_size, max_guide_diffs=a.max_guide_diffs,
                                  max_pam_mismatches=a.max_pam_mismatches, max_gaps_between_guide_and_pam=a.max_gaps_between_guide_and_pam,
                                  max_total_diffs=a.max_total_diffs, max_overlap=a.max_overlap,
                                  guide_mismatch_net_cost=a.guide_mismatch_net_cost, pam_mismatch_net_cost=a.pam_mismatch_net_cost,
                                  genome_gap_net_cost=a.genome_gap_net_cost, guide_gap_net_cost=a.guide_gap_net_cost, device=a.device)
        if a.output is None:
            sys.stdout.write(text)
    elif a.tool == "PairwiseAlignSequences":
        pairwise_align_sequences(a.input, a.output, max_gaps_between_guide_and_pam=a.max_gaps_between_guide_and_pam,
                                 mismatch_net_cost=a.guide_mismatch_net_cost, pam_mismatch_net_cost=a.pam_mismatch_net_cost,
                                 genome_gap_net_cost=a.genome_gap_net_cost, guide_gap_net_cost=a.guide_gap_net_cost, device=a.device)
    else:
        prepare_vcf(a.input, a.output, min_af=a.min_af, add_chr_prefix=a.add_chr_prefix, dict_path=a.dict)
    return 0


if __name__ == "__main__":
    sys.exit(main())
