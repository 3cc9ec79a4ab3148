#!/bin/bash
# Round 5, "before" evidence (HEAD of round 4) inside one gpurun call: bash tools/refresh_r05_before.sh
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench n1 $?"
python3 tools/trace_marks.py 1.0 6 2> $OUT/host_marks.txt > /dev/null; echo "marks $?"
bash tools/pmc_pass.sh tailwait "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" 1.0 3 hits > $OUT/pmc_tail_wait.txt 2>&1; echo "pmc tail $?"
bash tools/pmc_pass.sh tailwait2 "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" 1.0 3 hits > $OUT/pmc_tail_wait2.txt 2>&1; echo "pmc tail2 $?"
bash tools/prof_bench.sh --config 4 --steps 2 --warmup 1 > $OUT/prof_config4.txt 2>&1; cp gpurun_out/kernel_stats_bench.csv $OUT/kernel_stats_config4.csv; echo "prof c4 $?"
python3 tools/owned_speed.py > $OUT/owned_speed.txt 2>&1; echo "owned $?"
