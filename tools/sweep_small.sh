#!/bin/bash
for scale in 0.125 0.25 0.5; do
  for w in 1 1:1 3:2 2:1 1:1:1 2:1:1 3:2:1 5:3:2; do
    echo "scale $scale chunks $w"
    CALITAS_CHUNKS=$w python tools/ab_env.py CALITAS_BINNED 1 0 $scale 20 2>&1 | grep median || exit 1
  done
done
