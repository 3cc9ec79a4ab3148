#!/usr/bin/env python3
"""profiles/<tag>_tail_census.json: the vector / scalar instructions of every kernel BEHIND the scan per hg38-sized pass of guide #0, from one
counter pass of tools/pmc_pass.sh over tools/scan_profile.py 1.0 3 hits (CALITAS_CHUNKS=1: one launch of every kernel per pass).
bench.py reads it for roofline.tail.  python3 tools/tail_census.py gpurun_out/pmc_insts.csv STEPS OUT.json"""
import csv, json, sys


def main():
    path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    per = {}
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("calitas::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "rocprim" in n:
            n = "rocprim"
        if "scan_rows_kernel" in n or "planes_kernel" in n or "window_table" in n or "dpp_selftest" in n:
            continue
        d = per.setdefault(n, {"SQ_INSTS_VALU": 0.0, "SQ_INSTS_SALU": 0.0, "launches": 0})
        if r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
            d[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            d["launches"] += 1
    for d in per.values():
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
            d[k] = round(d[k] / steps)
        d["launches"] = round(d["launches"] / steps, 2)
    total = {"valu_wave_inst_per_pass": sum(d["SQ_INSTS_VALU"] for d in per.values()), "salu_wave_inst_per_pass": sum(d["SQ_INSTS_SALU"] for d in per.values())}
    res = {"workload": "hg38-sized synthetic genome of tools/scan_profile.py, guide #0, d = 5, one contig range (CALITAS_CHUNKS=1): the per-bin tail",
           "method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES (tools/pmc_pass.sh insts ... 1.0 %d hits), sums over a pass's launches / passes" % steps,
           "total": total, "kernels": dict(sorted(per.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]))}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["total"]), len(per), "kernels")


if __name__ == "__main__":
    main()
