#!/bin/bash
# Range weights x tail (general kernels / per-bin kernels) of the hg38-sized call (inside gpurun): bash tools/sweep_ranges.sh [scale]
scale=${1:-1.0}
for w in 5:3:2 6:3:1 4:3:2:1 5:3:1.5:0.5 3:2 2:1 3:1 1:1 4:4:2 5:4:1; do
  for b in 0 1 last; do
    echo "scale $scale chunks $w binned $b"
    CALITAS_CHUNKS=$w python tools/ab_env.py CALITAS_BINNED $b $b $scale 20 2>&1 | grep median | head -1 || exit 1
  done
done
