#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
python3 - /tmp/tl <<'PY'
import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    big = [r for r in rows if int(r.get("Bytes", 0) or 0) > (1 << 20)]
    for r in big[-6:]:
        print({k: r[k] for k in r})
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    long_copy = [r for r in rows if "rocclr" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
    print("rocclr kernels > 100 us:", len(long_copy))
    for r in long_copy[-6:]:
        print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us", "queue", r.get("Queue_Id"))
PY
