#!/bin/bash
# One rocprofv3 counter pass over tools/scan_profile.py (inside gpurun): bash tools/pmc_pass.sh NAME "COUNTER [COUNTER...]" [scale] [steps] [hits]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; COUNTERS=$2; SCALE=${3:-1.0}; STEPS=${4:-3}; HITS=${5:-}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$NAME
timeout -k 10 500 rocprofv3 --pmc $COUNTERS --output-format csv -d /tmp/pmc_$NAME -o run -- python3 "$ROOT/tools/scan_profile.py" $SCALE $STEPS $HITS > /tmp/pmc_$NAME.log 2>&1 < /dev/null
echo "exit $?"; tail -2 /tmp/pmc_$NAME.log
f=$(find /tmp/pmc_$NAME -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
  python3 - "$f" "$ROOT/gpurun_out/pmc_$NAME.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "calitas" in r.get("Kernel_Name", "")]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
agg = {}
for r in keep:
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:], r["Counter_Name"])
    agg.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-42s %-22s n=%d mean=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
fi
