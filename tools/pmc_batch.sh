#!/bin/bash
# Instruction census of a guide batch (inside gpurun): bash tools/pmc_batch.sh [guides] -> gpurun_out/pmc_batch.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-8}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_batch
timeout -k 10 700 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/pmc_batch -o run -- python3 "$ROOT/tools/batch_census.py" run $N > /tmp/pmc_batch.log 2>&1 < /dev/null
echo "exit $?"; cp /tmp/pmc_batch.log "$ROOT/gpurun_out/pmc_batch.log"; tail -3 /tmp/pmc_batch.log
f=$(find /tmp/pmc_batch -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 "$ROOT/tools/batch_census.py" sum "$f" $N > "$ROOT/gpurun_out/pmc_batch.txt" && cat "$ROOT/gpurun_out/pmc_batch.txt"
