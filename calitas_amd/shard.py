"""Multi-GPU partitioning of the SearchReference path (one process per GPU, no data-path collective).

Two natural partitions exist:
  * contigs -- every window x strand is independent (SearchReference.scala:537-561), and removeOverlaps
    (SearchReference.scala:653-675) groups by chromosome, so a rank that owns whole contigs can finish its rows alone;
    rank 0 only concatenates the per-contig row blocks in sequence-dictionary order (ReferenceHit.scala:284).  With
    consecutive ranges (contiguous_partition) that is a plain concatenation of the ranks' texts; LPT packing balances a
    little better (1.05x of the mean against 1.14x for hg38 on 8 ranks) but needs the blocks re-ordered.
  * guides  -- the reference runs one guide per invocation, so a 96-guide batch is 96 independent outputs.
  * windows -- the finest grain: windowIterator's sequence of windows (SearchReference.scala:39-71, contigs in order, starts
    0, step, 2 step, ...) cut into consecutive ranges of equal size, wherever that falls (window_partition).  A rank runs
    calitas_search on its range (calitas_params_t first_window / n_windows: every window is aligned exactly as in the whole job,
    the window's own overlap is its halo); contigs that lie entirely inside a range finish on their rank as before, and for a
    contig that is cut the alignments of its parts go to one rank -- in range order they are the alignments of the whole contig --
    which runs removeOverlaps / sort / rows on them (SearchReference.scala:641-675).  Balance: every rank gets the same number of
    windows, i.e. the same number of new bases to within one window step.
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


def contiguous_partition(lengths, n_bins):
    """Contigs in dictionary order cut into n_bins consecutive ranges with the smallest possible largest range (linear
    partition).  Returns a list of n_bins lists of contig indices (some empty when there are fewer contigs than bins).
    A rank that owns a consecutive range produces a consecutive piece of the final hits.txt: the gather is a concatenation."""
    n = len(lengths)
    k = max(1, min(n_bins, n))
    prefix = [0]
    for x in lengths:
        prefix.append(prefix[-1] + x)
    INF = float("inf")
    # best[j][i]: smallest largest range when the first i contigs form j ranges
    best = [[INF] * (n + 1) for _ in range(k + 1)]
    cut = [[0] * (n + 1) for _ in range(k + 1)]
    best[0][0] = 0
    for j in range(1, k + 1):
        for i in range(j, n + 1):
            for m in range(j - 1, i):
                cost = max(best[j - 1][m], prefix[i] - prefix[m])
                if cost < best[j][i]:
                    best[j][i], cut[j][i] = cost, m
    bounds, i = [], n
    for j in range(k, 0, -1):
        bounds.append((cut[j][i], i))
        i = cut[j][i]
    ranges = [list(range(a, b)) for a, b in reversed(bounds)]
    return ranges + [[] for _ in range(n_bins - k)]


def concat_rank_texts(texts):
    """hits.txt texts of consecutive contig ranges (bytes or str, each with the header line) -> the whole file."""
    nl = b"\n" if isinstance(texts[0], bytes) else "\n"
    out = [texts[0]]
    for t in texts[1:]:
        out.append(t[t.index(nl) + 1:])
    return (b"" if isinstance(texts[0], bytes) else "").join(out)


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


def window_counts(lengths, step):
    """|Range(0, len - 1, step)| per contig: the windows windowIterator makes (SearchReference.scala:52)."""
    return [0 if n < 2 else (n - 2) // step + 1 for n in lengths]


def window_partition(lengths, n_bins, step):
    """The job's windows (global index: contig-major, as calitas_window_table lists them) cut into n_bins consecutive ranges whose
    sizes differ by at most one.  Returns a list of (first_window, n_windows)."""
    total = sum(window_counts(lengths, step))
    out, first = [], 0
    for b in range(n_bins):
        n = total // n_bins + (1 if b < total % n_bins else 0)
        out.append((first, n))
        first += n
    return out


def range_contigs(lengths, step, first_window, n_windows):
    """What a window range covers: a list of (contig index, first window on the contig, number of windows, whole) in contig order,
    whole = the range holds every window of that contig."""
    out, base = [], 0
    for ci, nw in enumerate(window_counts(lengths, step)):
        a, b = max(first_window, base), min(first_window + n_windows, base + nw)
        if a < b:
            out.append((ci, a - base, b - a, b - a == nw))
        base += nw
    return out


def range_bases(lengths, step, window_size, first_window, n_windows):
    """Reference bases the windows of a range cover (the scan work of the rank that owns it)."""
    total = 0
    for ci, k0, n, _ in range_contigs(lengths, step, first_window, n_windows):
        total += min(lengths[ci], (k0 + n - 1) * step + window_size) - k0 * step
    return total


def contig_owner(parts_per_rank):
    """parts_per_rank[r] = range_contigs(...) of rank r.  {contig index: rank that finishes it} -- the lowest rank touching it."""
    owner = {}
    for r, parts in enumerate(parts_per_rank):
        for ci, _, _, _ in parts:
            owner.setdefault(ci, r)
    return owner


def owned_stretch(lengths, step, first_window, n_windows):
    """The stretch of the genome a window range OWNS under calitas_search_hits (include/calitas_hip.h, calitas_params_t.first_window):
    the hits whose coordinate_start lies at or behind the start of window first_window and before the start of window
    first_window + n_windows.  Returns ((contig, position), (contig, position)) -- a half-open interval of (contig, position) keys;
    the end of the reference is (number of contigs, 0).  Consecutive ranges own consecutive stretches, so their hits.txt texts
    concatenate (coordinate_start is the first key of ReferenceHit.sort)."""
    counts = window_counts(lengths, step)

    def start_of(w):
        base = 0
        for ci, nw in enumerate(counts):
            if w < base + nw:
                return (ci, (w - base) * step)
            base += nw
        return (len(lengths), 0)
    return start_of(first_window), start_of(first_window + n_windows)


def owns(stretch, contig, position):
    """Whether a hit at (contig index, coordinate_start) belongs to the stretch of owned_stretch()."""
    return stretch[0] <= (contig, position) < stretch[1]
