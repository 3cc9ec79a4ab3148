#!/bin/bash
# Round 5's profile set, inside one gpurun call: bash tools/refresh_r05.sh  (results under gpurun_out/r05f/, copied into profiles/ afterwards)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05f; mkdir -p $OUT
python3 bench.py > $OUT/r05_bench_n1.json 2> $OUT/bench_n1.err; echo "bench n1 $?"
CALITAS_CHUNKS=1 python3 bench.py --cpu-sample-mb 0 > $OUT/r05_bench_one_lane.json 2> $OUT/bench_one_lane.err; echo "one lane $?"
python3 tools/c2_speed.py > $OUT/r05_c2_speed.txt 2>&1; echo "c2 $?"
python3 tools/owned_speed.py > $OUT/r05_owned_speed.txt 2>&1; echo "owned $?"
python3 tools/trace_marks.py 1.0 6 2> $OUT/r05_host_marks.txt > /dev/null; echo "marks $?"
CALITAS_LANE_PRIO=low python3 tools/trace_marks.py 1.0 6 2> $OUT/r05_host_marks_lanes_low.txt > /dev/null; echo "marks low $?"
bash tools/prof_bench.sh --steps 20 --warmup 3 > $OUT/prof_bench.txt 2>&1; cp gpurun_out/kernel_stats_bench.csv $OUT/r05_rocprofv3_kernel_stats_bench.csv; echo "prof $?"
bash tools/prof_bench.sh --config 4 --steps 2 --warmup 1 > $OUT/prof_config4.txt 2>&1; cp gpurun_out/kernel_stats_bench.csv $OUT/r05_rocprofv3_kernel_stats_config4.csv; echo "prof c4 $?"
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $OUT/r05_timeline_lanes.txt; echo "timeline $?"
TIMELINE_MIN_COPY=0 bash tools/timeline.sh --scale 0.125 > /dev/null 2>&1; cp gpurun_out/timeline.txt $OUT/r05_timeline_slice8.txt; echo "timeline8 $?"
python3 bench.py --config 4 --steps 3 > $OUT/r05_bench_config4.json 2> $OUT/bench_config4.err; echo "config4 $?"
python3 bench.py --config 5 --steps 2 --warmup 1 > $OUT/r05_bench_config5_full.json 2> $OUT/bench_config5.err; echo "config5 $?"
bash tools/pmc_pass.sh fetch "FETCH_SIZE" 1.0 3 > $OUT/pmc_fetch.txt 2>&1; cp gpurun_out/pmc_fetch.csv $OUT/r05_pmc_fetch_size.csv; echo "pmc fetch $?"
bash tools/pmc_pass.sh write "WRITE_SIZE" 1.0 3 > $OUT/pmc_write.txt 2>&1; cp gpurun_out/pmc_write.csv $OUT/r05_pmc_write_size.csv; echo "pmc write $?"
bash tools/pmc_pass.sh insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" 1.0 3 > $OUT/pmc_insts.txt 2>&1; cp gpurun_out/pmc_insts.csv $OUT/r05_pmc_insts.csv; echo "pmc insts $?"
CALITAS_CHUNKS=1 bash tools/pmc_pass.sh tailinsts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" 1.0 3 hits > $OUT/r05_pmc_tail_insts.txt 2>&1; cp gpurun_out/pmc_tailinsts.csv $OUT/r05_pmc_tail_insts.csv; echo "pmc tail insts $?"
CALITAS_CHUNKS=1 bash tools/pmc_pass.sh tailwait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" 1.0 3 hits > $OUT/r05_pmc_tail_wait.txt 2>&1; echo "pmc tail wait $?"
python3 tools/tail_census.py $OUT/r05_pmc_tail_insts.csv 3 $OUT/r05_tail_census.json; echo "census $?"
bash tools/ab_r04.sh > $OUT/r05_ab_r04.txt 2>&1; echo "ab $?"
python3 bench.py --gpus 6 --scale 0.125 --rehearse-on-one-gpu --steps 3 --warmup 1 --cpu-sample-mb 0 > $OUT/r05_rehearse6.json 2> $OUT/rehearse6.err; echo "rehearse $?"
bash tools/cgroup_stat.sh > $OUT/r05_cgroup.txt 2>&1
