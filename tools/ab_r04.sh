#!/bin/bash
# Same-box A/B of the round-4 library (build/r04/libcalitas_hip.so, built from commit 09210ac) against the tree's: bash tools/ab_r04.sh
OUT=gpurun_out/r05_ab; mkdir -p $OUT
for i in 1 2; do
  CALITAS_LIB_PATH=$PWD/build/r04/libcalitas_hip.so python3 tools/owned_speed.py 2>&1 | tail -1 | sed "s/^/r04  /"
  python3 tools/owned_speed.py 2>&1 | tail -1 | sed "s/^/r05  /"
done
for i in 1 2; do
  CALITAS_LIB_PATH=$PWD/build/r04/libcalitas_hip.so python3 bench.py --config 4 --steps 3 --cpu-sample-mb 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r04  config4', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items() if v})"
  python3 bench.py --config 4 --steps 3 --cpu-sample-mb 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05  config4', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items() if v})"
done
for i in 1 2; do
  CALITAS_LIB_PATH=$PWD/build/r04/libcalitas_hip.so python3 bench.py --cpu-sample-mb 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r04  config3', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernel_ms'].items() if v})"
  python3 bench.py --cpu-sample-mb 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05  config3', round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['kernel_ms'].items() if v})"
done
