#!/usr/bin/env python3
"""Instruction census of scan_rows_kernel from the compiler's own output, priced with the measured per-instruction issue times: what
bench.py's roofline.valu.mix_limit is computed from (so that the figure follows the kernel, not a comment).

  python tools/scan_census.py [--nw 8] [--nwarm 1] [-o profiles/r04_scan_census.json]

1. `hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S calitas_amd/csrc/scan_rows.hip` (the flags of the Makefile).
2. The instantiation scan_rows_kernel<NW, NWARM>: its basic blocks with LLVM's loop annotations.  Per strand the kernel is a loop over
   the guides of the launch (depth 1) around the loop over the protospacer rows (depth 2).  Vector-ALU instructions are counted per
   mnemonic for
     * one iteration of the row loop -- blocks of the loop that differ only in operands are the alternatives of the switch on the
       row's base (A / C / G / T / a set) and count once;
     * the rest of one guide x strand (the bottom-row test of the lane's words, staging, the reverse strand's bit reversal) -- without
       the blocks behind an `s_cbranch_execz` (suspect words, record appends: rare, data dependent).
3. Issue time per mnemonic: profiles/valu_rates.json (tools/valu_bench*.hip on an MI355X, profiles/r01_valu_rates.txt), ns per
   wave-instruction per SIMD; a mnemonic without a measurement takes the rate of its class (VOP3 with three sources and most two-operand
   VOP3-only encodings issue at half rate on gfx950).
The output holds the census, the rates used and the two times bench.py needs: ns per row iteration and ns per guide x strand outside
the row loop, per wave."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm():
    src = os.path.join(ROOT, "calitas_amd", "csrc", "scan_rows.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "scan_rows.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_body(asm, nw, nwarm):
    sym = "_ZN7calitas12_GLOBAL__N_116scan_rows_kernelILi%dELi%dEEEvNS_8ScanArgsE" % (nw, nwarm)
    start = asm.index("\n" + sym + ":")
    end = asm.index("s_endpgm", start)
    return sym, asm[start:end].splitlines()


def blocks_of(lines):
    """Basic blocks in layout order: {label, depth, header, ops, succ}.  A block starts at a label (.LBBk_n:) or at the comment the
    compiler leaves for an unlabelled one (; %bb.n:); depth / loop header come from LLVM's loop annotations on those lines."""
    out, cur = [], None
    for i, ln in enumerate(lines):
        m = re.match(r"^\.L(BB\d+_\d+):\s*(;.*)?$", ln) or re.match(r"^; %(bb\.\d+):\s*(;.*)?$", ln)
        if m:
            label, note = m.group(1), m.group(2) or ""
            for more in lines[i + 1:i + 4]:                     # (the annotation of a loop header continues on comment-only lines)
                if more.strip().startswith(";") and "%bb." not in more:
                    note += " " + more.strip()
                else:
                    break
            depth, header = 0, None
            h = re.search(r"This Inner Loop Header: Depth=(\d+)|This Loop Header: Depth=(\d+)", note)
            if h:
                depth, header = int(h.group(1) or h.group(2)), label
            else:
                h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
                if h:
                    header, depth = h.group(1), int(h.group(2))
            parents = re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", note)
            cur = {"label": label, "depth": depth, "header": header, "ops": [], "branches": [], "parents": [p for p, _ in parents]}
            out.append(cur)
            continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        f = t.split()
        if cur is None:
            cur = {"label": "entry", "depth": 0, "header": None, "ops": [], "branches": [], "parents": []}
            out.append(cur)
        cur["ops"].append(f[0])
        if f[0].startswith("s_cbranch") or f[0] == "s_branch":
            cur["branches"].append((f[0], f[1].lstrip(".L") if len(f) > 1 else None))
    for k, b in enumerate(out):                               # successors: branch targets, and the next block unless the block ends in s_branch
        succ = [t for _, t in b["branches"] if t]
        if not (b["ops"] and b["ops"][-1] == "s_branch") and not (b["ops"] and b["ops"][-1] == "s_endpgm") and k + 1 < len(out):
            succ.append(out[k + 1]["label"])
        b["succ"] = succ
    return out


def all_paths(members, header, limit=4096):
    """Every simple way once around the loop `header` through `members`: lists of labels."""
    out = []

    def walk(label, path):
        if len(out) >= limit:
            return
        for nx in members[label]["succ"]:
            if nx == header:
                out.append(list(path))
            elif nx in members and nx not in path:
                walk(nx, path + [nx])
    walk(header, [header])
    return out


def cheapest_path(blocks, members, header, cost_of):
    """Cheapest way once around the loop `header` through the blocks `members` (label -> block): from the header along successors
    inside the loop until an edge returns to the header.  Returns (cost, [labels])."""
    best = [None, None]

    def walk(label, cost, path):
        if best[0] is not None and cost >= best[0]:
            return
        for nx in members[label]["succ"]:
            if nx == header:
                if best[0] is None or cost < best[0]:
                    best[0], best[1] = cost, list(path)
            elif nx in members and nx not in path:
                walk(nx, cost + cost_of(members[nx]), path + [nx])
    walk(header, cost_of(members[header]), [header])
    return best[0], best[1]


def census(ops):
    c = {}
    for op in ops:
        if op.startswith("v_"):
            key = re.sub(r"_e(32|64)$", "", op)
            c[key] = c.get(key, 0) + 1
    return c


def add(a, b, w=1):
    for k, v in b.items():
        a[k] = a.get(k, 0) + v * w
    return a


def rate_of(op, rates):
    if op in rates["ns_per_wave_inst_per_simd"]:
        return rates["ns_per_wave_inst_per_simd"][op]
    for pat, cls in rates["classes"]:
        if re.match(pat, op):
            return rates["class_ns"][cls]
    return rates["class_ns"]["full"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nw", type=int, default=8)
    ap.add_argument("--nwarm", type=int, default=1)
    ap.add_argument("-o", "--output", default=os.path.join(ROOT, "profiles", "r04_scan_census.json"))
    a = ap.parse_args()
    rates = json.load(open(os.path.join(ROOT, "profiles", "valu_rates.json")))
    sym, lines = kernel_body(device_asm(), a.nw, a.nwarm)
    blocks = blocks_of(lines)
    # the two row loops (forward strand first): depth-2 headers; their guide loops: the depth-1 loop they sit in
    d2 = [b for b in blocks if b["depth"] == 2 and b["header"] == b["label"]]
    if len(d2) < 2:
        sys.exit("scan_census: expected a row loop per strand in %s, found %d depth-2 loops" % (sym, len(d2)))

    def ns_of(c):
        return sum(n * rate_of(op, rates) for op, n in c.items())
    strands = []
    for hb in d2:
        h2 = hb["label"]
        inner = {b["label"]: b for b in blocks if b["depth"] == 2 and b["header"] == h2}
        # one iteration = the cheapest way around the loop: the row's base selects one of the alternatives (A / C / G / T, or a set of
        # them for an IUPAC code -- dearer, and not what a 20-mer of plain bases runs)
        # (chosen as the census most of the ways around the loop share: the four plain bases compile to the same instructions with other
        # truth tables; the way that skips every alternative exists in the flow graph only)
        by_census = {}
        for pth in all_paths(inner, h2):
            c = {}
            for lb in pth:
                add(c, census(inner[lb]["ops"]))
            by_census.setdefault(tuple(sorted(c.items())), []).append(pth)
        sig = max(by_census, key=lambda k: (len(by_census[k]), -ns_of(dict(k))))
        path, row = by_census[sig][0], dict(sig)
        alternatives = {"ways_around_the_loop": sum(len(v) for v in by_census.values()), "with_this_census": len(by_census[sig]),
                        "other_censuses_valu": sorted(sum(n for _, n in k) for k in by_census if k != sig)}
        h1 = hb["parents"][0] if hb["parents"] else None
        outer = {b["label"]: b for b in blocks if b["depth"] == 1 and b["header"] == h1}
        # the guide loop around it, the row loop taken as one step (it is priced per row): again the cheapest way around -- the blocks
        # behind `s_cbranch_execz` (a suspect word, a record to append) are data dependent and rare
        first_inner = [lb for lb in inner]
        for b in outer.values():
            b["succ"] = [("@rows" if sx in inner else sx) for sx in b["succ"]]
        exits = set()
        for b in inner.values():
            for sx in b["succ"]:
                if sx in outer:
                    exits.add(sx)
        outer["@rows"] = {"label": "@rows", "ops": [], "succ": sorted(exits)}
        _, opath = cheapest_path(blocks, outer, h1, lambda b: ns_of(census(b["ops"])))
        rest = {}
        for lb in opath or []:
            add(rest, census(outer[lb]["ops"]))
        strands.append({"row_loop_header": h2, "guide_loop_header": h1, "row_iteration_path": path, "row_iteration": row, "row_iteration_alternatives": alternatives,
                        "guide_strand_path": opath, "guide_strand_rest": rest, "row_loop_blocks": first_inner})
    # the kernel holds the row loops twice per strand: for plain tiles and for tiles with exception bases in the text (N runs' edges,
    # contig ends, IUPAC codes: 392 of 23 616 tiles of the hg38-sized genome), whose Eq needs a third plane.  Priced: the plain pair.
    strands.sort(key=lambda s_: ns_of(s_["row_iteration"]))
    masked = strands[2:]
    strands = strands[:2]
    words = a.nw + a.nwarm
    once = {}
    for b in blocks:                                          # outside every loop: staging, the reverse strand's planes, the flush of the records
        if b["depth"] == 0:
            add(once, census(b["ops"]))

    ns = ns_of
    out = {"kernel": sym, "source": "hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S calitas_amd/csrc/scan_rows.hip (tools/scan_census.py)",
           "words_per_lane": words, "strands": strands, "rates": rates,
           "row_iteration_ns": [round(ns(s["row_iteration"]), 2) for s in strands],
           "ns_per_word_row": [round(ns(s["row_iteration"]) / words, 3) for s in strands],
           "guide_strand_rest_ns": [round(ns(s["guide_strand_rest"]), 2) for s in strands],
           "once_per_wave": once, "once_per_wave_ns": round(ns(once), 2),
           "tiles_with_exception_bases": {"row_iteration_ns": [round(ns(s["row_iteration"]), 2) for s in masked],
                                          "valu_per_row_iteration": [sum(s["row_iteration"].values()) for s in masked]},
           "valu_per_row_iteration": [sum(s["row_iteration"].values()) for s in strands]}
    with open(a.output, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps({k: out[k] for k in ("kernel", "words_per_lane", "row_iteration_ns", "ns_per_word_row", "guide_strand_rest_ns", "once_per_wave_ns", "valu_per_row_iteration")}, indent=1))
    for s in strands:
        print(s["row_loop_header"], "row iteration:", dict(sorted(s["row_iteration"].items(), key=lambda kv: -kv[1])))
        print("   rest:", dict(sorted(s["guide_strand_rest"].items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
