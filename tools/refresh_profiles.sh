#!/bin/bash
# Regenerates the round's files for profiles/ on the GPU box (inside gpurun): bash tools/refresh_profiles.sh r02
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r02}
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/bench_n1.err"; echo "bench n1 $?"
CALITAS_CHUNKS=1 python3 bench.py --cpu-sample-mb 0 > "$OUT/${TAG}_bench_one_lane.json" 2> "$OUT/bench_one_lane.err"; echo "bench one lane $?"
python3 bench.py --config 4 --steps 3 > "$OUT/${TAG}_bench_config4.json" 2> "$OUT/bench_config4.err"; echo "bench config4 $?"
python3 bench.py --config 5 > "$OUT/${TAG}_bench_config5.json" 2> "$OUT/bench_config5.err"; echo "bench config5 $?"
CALITAS_TRACE=1 STRESS_STREAM=1 python3 tools/stress_c5.py 1.0 8 2 > "$OUT/${TAG}_stress_c5_stream.txt" 2>&1; echo "stress stream $?"
CALITAS_TRACE=1 python3 tools/c5_phases.py 1.0 > "$OUT/${TAG}_stress_c5_block.txt" 2>&1; echo "stress block $?"
python3 tools/c2_speed.py > "$OUT/${TAG}_c2_speed.txt" 2>&1; echo "c2 $?"
python3 tools/trace_marks.py 1.0 3 2> "$OUT/${TAG}_host_marks.txt" > /dev/null; echo "marks $?"
bash tools/prof_bench.sh --steps 20 --warmup 3 > "$OUT/prof_bench.txt" 2>&1; cp gpurun_out/kernel_stats_bench.csv "$OUT/${TAG}_rocprofv3_kernel_stats_bench.csv"; echo "prof $?"
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt "$OUT/${TAG}_timeline_lanes.txt"; echo "timeline $?"
TIMELINE_MIN_COPY=0 TIMELINE_PROG=tools/c2_speed.py bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt "$OUT/${TAG}_timeline_config2_call.txt"; echo "timeline c2 $?"
CALITAS_CHUNKS=1 bash tools/pmc_pass.sh tail "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" 1.0 2 1 > "$OUT/${TAG}_pmc_tail_kernels.txt" 2>&1; echo "pmc $?"
