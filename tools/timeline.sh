#!/bin/bash
# Kernel + copy timeline of the last calitas_search_hits call of a short bench run (inside gpurun): bash tools/timeline.sh [bench args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
# TIMELINE_PROG=tools/c2_speed.py: another driver than bench.py (the timeline then starts at its last scan kernel)
if [ -n "$TIMELINE_PROG" ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o run -- python3 "$ROOT/$TIMELINE_PROG" "$@" > /tmp/tl.log 2>&1 < /dev/null
else
  timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o run -- python3 "$ROOT/bench.py" --cpu-sample-mb 0 --steps 3 --warmup 1 "$@" > /tmp/tl.log 2>&1 < /dev/null
fi
python3 - /tmp/tl "$ROOT/gpurun_out/timeline.txt" <<'PY'
import csv, glob, sys
d, out = sys.argv[1], sys.argv[2]
import os
MIN_COPY = int(os.environ.get("TIMELINE_MIN_COPY", str(1 << 20)))   # bytes; 0 lists every copy
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "calitas" not in n and "ROCPRIM_400200" not in n and "__amd_rocclr" not in n:
            continue
        n = n.replace("calitas::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if "rocprim" in n:
            n = "rocprim:" + ("merge" if "merge_sort_block_merge" in n else "blocksort" if "block_sort" in n else "scan" if "scan" in n else "other")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), n[-28:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        b = int(r.get("Bytes", r.get("bytes", 0)) or 0)
        if b < MIN_COPY:
            continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", "%s %.1f MB" % (r.get("Direction", ""), b / 1e6)))
ev.sort()
# the last call starts at the last-but-one scan_kernel launch group: find the last two scan kernels
scans = [e for e in ev if "scan_kernel" in e[3] or "scan_rows_kernel" in e[3]]
if not ev:
    print(open("/tmp/tl.log").read()[-3000:]); sys.exit(1)
t0 = scans[-2][0] if len(scans) >= 2 else ev[0][0]
if os.environ.get("TIMELINE_PROG"):
    t0 = scans[-1][0] if scans else ev[0][0]
with open(out, "w") as f:
    for s, e, q, n in ev:
        if s < t0:
            continue
        f.write("%9.1f %9.1f  %7.1f us  %-6s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
print(open(out).read())
PY
