#!/bin/bash
# alternate two builds of the library, whole processes: bash ab_lib.sh LIB_B [scale] [rounds]
B=$1; scale=${2:-1.0}; rounds=${3:-4}
for i in $(seq $rounds); do
  echo -n "A "; python tools/ab_env.py CALITAS_NOTHING - - $scale 15 2>&1 | grep median | head -1
  echo -n "B "; CALITAS_LIB_PATH=$B python tools/ab_env.py CALITAS_NOTHING - - $scale 15 2>&1 | grep median | head -1
done
