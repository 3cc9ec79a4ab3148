#!/bin/bash
# Grid sweeps of align_kernel (x jobs per wave) and trace_kernel inside whole calitas_search_hits calls, on slices of the bench genome
# (inside gpurun): bash tools/sweep_align.sh > gpurun_out/sweep_align.txt   (profiles/r03_sweep_align.txt)
for scale in 1.0 0.25 0.125; do
  for lpj in 21 32; do
    for blocks in 96 128 192 256 384 512 683 896 1024; do
      echo "scale $scale lpj $lpj blocks $blocks"
      CALITAS_ALIGN_LPJ=$lpj CALITAS_ALIGN_BLOCKS=$blocks python tools/ab_env.py CALITAS_ALIGN_BLOCKS_NARROW $blocks $blocks $scale 20 2>&1 | grep median | head -1 || exit 1
    done
  done
done
for blocks in 16 64 256 1024; do
  echo "ecoli lpj 21 blocks $blocks"
  CALITAS_ALIGN_BLOCKS=$blocks CALITAS_ALIGN_BLOCKS_NARROW=$blocks python tools/c2_speed.py 2>&1 | grep "config 2" || exit 1
done
for scale in 1.0 0.125; do
  for tb in 256 512 1024 2048; do
    echo "scale $scale align 512 trace blocks $tb"
    CALITAS_ALIGN_BLOCKS=512 CALITAS_ALIGN_BLOCKS_NARROW=512 CALITAS_TRACE_BLOCKS=$tb python tools/ab_env.py CALITAS_TRACE_BLOCKS_NARROW $tb $tb $scale 20 2>&1 | grep median | head -1 || exit 1
  done
done
