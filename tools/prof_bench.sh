#!/bin/bash
# Kernel-trace statistics of `python3 bench.py <args>` on the GPU box; prints this library's kernels and keeps the full
# CSV under gpurun_out/.  Usage (inside gpurun): bash tools/prof_bench.sh [bench args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python3 "$ROOT/bench.py" --cpu-sample-mb 0 "$@" > /tmp/prof_bench.log 2>&1 < /dev/null
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/kernel_stats_bench.csv" \;
python3 - "$ROOT/gpurun_out/kernel_stats_bench.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "calitas" in n or "ROCPRIM_400200" in n:
        n = n.replace("calitas::", "").replace("(anonymous namespace)::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "")
        print("%-64s calls %4s avg %9.1f us" % (n[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 /tmp/prof_bench.log | cut -c1-200
