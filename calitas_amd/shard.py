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


# ---- where a rank's host threads run -------------------------------------------------------------------------------------
# A rank's call is bound by host round trips once its slice is small (an eighth of a genome: 0.5 ms, five dependent launches), and a
# round trip from the far socket costs a quarter more.  The reference pins nothing (one JVM, one pool:
# SearchReference.scala:75-94); a job of one process per GPU has to.

def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cpulist(text):
    out = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes(root="/"):
    """NUMA node of every GPU in HIP's device order: the KFD topology lists the GPUs in that order (nodes with simd_count > 0), each
    with the minor of its DRM render node, whose PCI device says which NUMA node it hangs off.  [] when the machine has no KFD
    topology (no GPU), None entries where the kernel does not say (-1)."""
    import os
    top = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    try:
        ids = sorted(int(d) for d in os.listdir(top) if d.isdigit())
    except OSError:
        return []
    out = []
    for i in ids:
        props = dict(ln.split(None, 1) for ln in (_read(os.path.join(top, str(i), "properties")) or "").splitlines() if " " in ln)
        if int(props.get("simd_count", "0")) <= 0:
            continue
        node = _read(os.path.join(root, "sys/class/drm/renderD%s/device/numa_node" % props.get("drm_render_minor", "-1")))
        out.append(int(node) if node is not None and int(node) >= 0 else None)
    return out


def rank_cpus(local_rank, n_local, allowed, root="/", visible=None, min_cpus=2):
    """The CPUs rank `local_rank` of `n_local` ranks on this machine should run on: those of its GPU's NUMA node that the process may
    use at all (`allowed`: os.sched_getaffinity), shared out evenly among the ranks whose GPUs hang off the same node.  `visible`: the
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES list when one is set (rank r then uses GPU visible[r]).  None: leave the process alone
    (no topology, the node unknown, or fewer than min_cpus CPUs for a rank)."""
    import os
    nodes = gpu_numa_nodes(root)
    gpu_of = lambda r: (visible[r] if visible and r < len(visible) else r)
    if not nodes or gpu_of(local_rank) >= len(nodes) or nodes[gpu_of(local_rank)] is None:
        return None
    mine = nodes[gpu_of(local_rank)]
    cpus = [c for c in _cpulist(_read(os.path.join(root, "sys/devices/system/node/node%d/cpulist" % mine))) if c in allowed]
    peers = [r for r in range(n_local) if gpu_of(r) < len(nodes) and nodes[gpu_of(r)] == mine]
    if local_rank not in peers:
        return None
    per = len(cpus) // len(peers)
    if per < min_cpus:
        return None
    k = peers.index(local_rank)
    return cpus[k * per:(k + 1) * per]


def pieces_match(whole, pieces, rows_whole=None):
    """Whether the ranks' texts of one guide are the whole job's text cut into consecutive pieces: `whole` = the bytes of a
    single-process hits.txt, pieces = [(crc32, n_bytes, n_rows) of rank r's text, header line included] in rank order.  A rank's text
    must be the header plus its consecutive piece of the body -- the ranks own consecutive stretches of the genome and coordinate_start
    is the first sort key (ReferenceHit.scala:284).  What bench.py's batch_sharded block checks for a sample of guides."""
    import zlib
    head = whole[:whole.index(b"\n") + 1]
    off, rows = len(head), 0
    for crc, nbytes, nrows in pieces:
        body = int(nbytes) - len(head)
        if body < 0 or zlib.crc32(head + whole[off:off + body]) != int(crc):
            return False
        off += body
        rows += int(nrows)
    return off == len(whole) and (rows_whole is None or rows == rows_whole)


def pin_rank(local_rank, n_local, root=None):
    """sched_setaffinity for this process (before any thread pool exists) per rank_cpus; returns the CPU list or None.
    root (or CALITAS_BENCH_SYSFS_ROOT): another tree than / to read the topology from (tests)."""
    import os
    root = root or os.environ.get("CALITAS_BENCH_SYSFS_ROOT") or "/"
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    try:
        visible = [int(x) for x in vis.split(",")] if vis else None
    except ValueError:
        visible = None                                         # (UUIDs: no way to map them here)
        if vis:
            return None
    try:
        cpus = rank_cpus(local_rank, n_local, os.sched_getaffinity(0), root=root, visible=visible)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, AttributeError, ValueError):
        return None


def cgroup_cpu_quota(root=None):
    """The CPU quota of this process's cgroup in cores (cpu.max: quota / period; the tightest one on the way up the hierarchy), or None
    when there is none.  A GPU box bounds a process by such a quota on a host it shares (nproc says 256, cpu.max says 16 cores'
    worth: DESIGN.md 4.7) -- affinity does not show it.  root (or CALITAS_BENCH_SYSFS_ROOT): another tree than / (tests)."""
    import os
    root = root or os.environ.get("CALITAS_BENCH_SYSFS_ROOT") or "/"
    try:
        with open(os.path.join(root, "proc/self/cgroup")) as f:
            lines = [ln.strip() for ln in f if ln.strip()]
    except OSError:
        return None
    best = None
    for ln in lines:
        parts = ln.split(":", 2)
        if len(parts) != 3 or parts[1] not in ("", "cpu", "cpu,cpuacct", "cpuacct,cpu"):
            continue
        d = parts[2].strip("/")
        while True:                                            # this cgroup and every one above it
            base = os.path.join(root, "sys/fs/cgroup", d) if d else os.path.join(root, "sys/fs/cgroup")
            for fn in ("cpu.max", "cpu/cpu.max"):
                try:
                    with open(os.path.join(base, fn)) as f:
                        q, _, per = f.read().strip().partition(" ")
                    if q != "max" and float(per or 100000) > 0:
                        cores = float(q) / float(per or 100000)
                        best = cores if best is None else min(best, cores)
                except (OSError, ValueError):
                    pass
            if not d:
                break
            d = os.path.dirname(d)
    return best


def worker_threads(cpus, n_local, quota=None, cap=16):
    """Worker threads of one rank's library pool: what its CPU list allows, and no more than its share of the box's CPU quota --
    eight ranks that each start sixteen workers on a sixteen-core quota throttle each other (a call keeps ~8 cores busy while its
    text is expanded).  At least 2."""
    n = len(cpus) if cpus else cap
    if quota:
        n = min(n, int(quota // max(1, n_local)))
    return max(2, min(cap, n))


def resident_contigs(lengths, step, first_window, n_windows):
    """The contigs a process of a multi-GPU job has to hold for its window range: those the range has a window on (the halo of a
    stretch -- a bin on either side -- never leaves a contig).  The others are given to calitas_set_reference without bases."""
    return [ci for ci, _, _, _ in range_contigs(lengths, step, first_window, n_windows)]
