#!/bin/bash
# ms_per_step of bench.py for several CALITAS_CHUNKS settings (run inside gpurun).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for c in "$@"; do
  CALITAS_CHUNKS=$c timeout -k 10 200 python3 bench.py --steps 12 --warmup 4 --cpu-sample-mb 0 2>/dev/null < /dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read()); print('chunks %-14s ms_per_step %.3f  scan %.3f align %.3f hitsk %.3f copy %.3f' % ('$c', b['ms_per_step'], b['kernel_ms']['scan'], b['kernel_ms']['align'], b['kernel_ms']['hits_kernels'], b['kernel_ms']['text_copy']))"
done
