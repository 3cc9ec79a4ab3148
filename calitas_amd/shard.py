"""Multi-GPU partitioning of the SearchReference path (one process per GPU, no data-path collective).

Two natural partitions exist:
  * contigs -- every window x strand is independent (SearchReference.scala:537-561), and removeOverlaps
    (SearchReference.scala:653-675) groups by chromosome, so a rank that owns whole contigs can finish its rows alone;
    rank 0 only concatenates the per-contig row blocks in sequence-dictionary order (ReferenceHit.scala:284).
  * guides  -- the reference runs one guide per invocation, so a 96-guide batch is 96 independent outputs.
"""


def lpt_partition(lengths, n_bins):
    """Longest-processing-time bin packing of contigs by length. Returns a list of n_bins lists of contig indices
    (each list in ascending contig order)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    loads = [0] * n_bins
    bins = [[] for _ in range(n_bins)]
    for i in order:
        b = min(range(n_bins), key=lambda k: (loads[k], k))
        bins[b].append(i)
        loads[b] += lengths[i]
    return [sorted(b) for b in bins]


def split_rows_by_contig(tsv_text, contig_names):
    """hits.txt text -> (header, {global contig index: [row lines]}) using the chromosome column."""
    lines = tsv_text.splitlines()
    header = lines[0]
    col = header.split("\t").index("chromosome")
    index = {n: i for i, n in enumerate(contig_names)}
    out = {}
    for ln in lines[1:]:
        out.setdefault(index[ln.split("\t")[col]], []).append(ln)
    return header, out


def merge_contig_rows(header, per_rank_blocks):
    """per_rank_blocks: list (one per rank) of {global contig index: [row lines]}. Rows of one contig come from exactly
    one rank and are already in final order; the merged file lists contigs in dictionary order."""
    merged = {}
    for blocks in per_rank_blocks:
        for ci, rows in blocks.items():
            if ci in merged:
                raise ValueError("contig %d reported by two ranks" % ci)
            merged[ci] = rows
    out = [header]
    for ci in sorted(merged):
        out.extend(merged[ci])
    return "\n".join(out) + "\n"


def guides_for_rank(n_guides, rank, world):
    """Round-robin assignment of guide indices to ranks."""
    return list(range(rank, n_guides, world))
