#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03f; mkdir -p $OUT
python3 bench.py > $OUT/r03_bench_n1.json 2> $OUT/bench_n1.err; echo "bench n1 $?"
CALITAS_CHUNKS=1 python3 bench.py --cpu-sample-mb 0 > $OUT/r03_bench_one_lane.json 2> $OUT/bench_one_lane.err; echo "one lane $?"
python3 tools/slice_speed.py 30 > $OUT/r03_slice_speed.txt 2>&1; echo "slice $?"
python3 tools/c2_speed.py > $OUT/r03_c2_speed.txt 2>&1; echo "c2 $?"
python3 tools/trace_marks.py 1.0 6 2> $OUT/r03_host_marks.txt > /dev/null; echo "marks $?"
bash tools/prof_bench.sh --steps 20 --warmup 3 > $OUT/prof_bench.txt 2>&1; cp gpurun_out/kernel_stats_bench.csv $OUT/r03_rocprofv3_kernel_stats_bench.csv; echo "prof $?"
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $OUT/r03_timeline_lanes.txt; echo "timeline $?"
TIMELINE_MIN_COPY=0 bash tools/timeline.sh --scale 0.125 > /dev/null 2>&1; cp gpurun_out/timeline.txt $OUT/r03_timeline_slice8.txt; echo "timeline slice $?"
TIMELINE_MIN_COPY=0 TIMELINE_PROG=tools/c2_speed.py bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $OUT/r03_timeline_config2_call.txt; echo "timeline c2 $?"
python3 tools/scan_interference.py > $OUT/r03_scan_interference.txt 2>&1; echo "interference $?"
python3 bench.py --config 4 --steps 3 > $OUT/r03_bench_config4.json 2> $OUT/bench_config4.err; echo "config4 $?"
