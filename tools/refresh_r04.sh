#!/bin/bash
# Round 4's profile set, inside one gpurun call: bash tools/refresh_r04.sh  (results under gpurun_out/r04f/, copied into profiles/ afterwards)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04f; mkdir -p $OUT
python3 bench.py > $OUT/r04_bench_n1.json 2> $OUT/bench_n1.err; echo "bench n1 $?"
CALITAS_CHUNKS=1 python3 bench.py --cpu-sample-mb 0 > $OUT/r04_bench_one_lane.json 2> $OUT/bench_one_lane.err; echo "one lane $?"
python3 tools/c2_speed.py > $OUT/r04_c2_speed.txt 2>&1; echo "c2 $?"
python3 tools/owned_speed.py > $OUT/r04_owned_speed.txt 2>&1; echo "owned $?"
python3 tools/trace_marks.py 1.0 6 2> $OUT/r04_host_marks.txt > /dev/null; echo "marks $?"
bash tools/prof_bench.sh --steps 20 --warmup 3 > $OUT/prof_bench.txt 2>&1; cp gpurun_out/kernel_stats_bench.csv $OUT/r04_rocprofv3_kernel_stats_bench.csv; echo "prof $?"
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $OUT/r04_timeline_lanes.txt; echo "timeline $?"
python3 bench.py --config 4 --steps 3 > $OUT/r04_bench_config4.json 2> $OUT/bench_config4.err; echo "config4 $?"
bash tools/pmc_pass.sh fetch "FETCH_SIZE" 1.0 3 > $OUT/pmc_fetch.txt 2>&1; cp gpurun_out/pmc_fetch.csv $OUT/r04_pmc_fetch_size.csv; echo "pmc fetch $?"
bash tools/pmc_pass.sh write "WRITE_SIZE" 1.0 3 > $OUT/pmc_write.txt 2>&1; cp gpurun_out/pmc_write.csv $OUT/r04_pmc_write_size.csv; echo "pmc write $?"
bash tools/pmc_pass.sh insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" 1.0 3 > $OUT/pmc_insts.txt 2>&1; cp gpurun_out/pmc_insts.csv $OUT/r04_pmc_insts.csv; echo "pmc insts $?"
python3 tools/slow_calls.py 600 > $OUT/slow_calls_now.txt 2>&1; echo "slow calls $?"
python3 tools/batch_probe.py 96 2 > $OUT/r04_batch_probe.txt 2>&1; echo "batch probe $?"
python3 tools/expand_speed.py 118000 30 > $OUT/r04_expand_speed.txt 2>&1; echo "expand speed $?"
bash tools/cgroup_stat.sh > $OUT/r04_cgroup.txt 2>&1
