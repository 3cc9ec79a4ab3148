#!/bin/bash
# Kernel-trace statistics of any driver under tools/ on the GPU box: bash tools/prof_prog.sh tools/stress_c5.py 0.05 8 1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
PROG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profp -o run -- python3 "$ROOT/$PROG" "$@" > /tmp/prof_prog.log 2>&1 < /dev/null
find /tmp/profp -name "*kernel_stats.csv" -exec cp {} "$ROOT/gpurun_out/kernel_stats_prog.csv" \;
python3 - "$ROOT/gpurun_out/kernel_stats_prog.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "calitas" in n or "ROCPRIM_400200" in n:
        n = n.replace("calitas::", "").replace("(anonymous namespace)::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "")
        print("%-64s calls %4s avg %9.1f us total %9.1f ms" % (n[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 /tmp/prof_prog.log | cut -c1-300
