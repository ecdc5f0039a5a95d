#!/bin/bash
# Everything under profiles/r02_* comes from this script (run on the GPU box: gpurun -- tools/gpu_profiles.sh).
set -u
o=gpurun_out/r02
mkdir -p $o
python bench.py --steps 20 --warmup 3 > $o/bench_line.json 2> $o/bench_line.err
python bench.py --steps 10 --warmup 2 --height 544 --width 960 --iters 32 --batch 1 --pyramid fp16 --no-cpu-baseline > $o/bench_c5_fp16_line.json 2> $o/bench_c5_fp16.err
python bench.py --steps 10 --warmup 2 --height 544 --width 960 --iters 32 --batch 1 --pyramid fp32 --no-cpu-baseline > $o/bench_c5_fp32_line.json 2> $o/bench_c5_fp32.err
python bench.py --steps 10 --warmup 2 --height 544 --width 960 --iters 32 --batch 4 --pyramid fp16 --no-cpu-baseline > $o/bench_c5_fp16_b4_line.json 2> $o/bench_c5_fp16_b4.err
python bench.py --mode train --steps 5 --warmup 2 > $o/train_line.json 2> $o/train_line.err
python tools/bench_pwc.py > $o/pwc_line.json 2> $o/pwc_line.err
tools/prof_trace.sh r02/bench_b8 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
tools/prof_trace.sh r02/train_b8 bench.py --mode train --steps 3 --warmup 1
ONLY=lookup tools/prof_pmc.sh r02/lookup_pmc tools/bench_lookup.py > $o/pmc_lookup.log 2>&1
HALF=1 ONLY=lookup tools/prof_pmc.sh r02/lookup_pmc_fp16 tools/bench_lookup.py > $o/pmc_lookup16.log 2>&1
ONLY=build tools/prof_pmc.sh r02/build_pmc tools/bench_lookup.py > $o/pmc_build.log 2>&1
python tools/make_traffic_json.py $o/lookup_pmc lookup_tiled_kernel 24576 2904 $o/lookup_traffic.json "B=8, 48x64 queries, random coords +-8 px (tools/bench_lookup.py), fp32 tiled pyramid, cache-warm and cache-cold launches"
python tools/make_traffic_json.py $o/lookup_pmc_fp16 lookup_tiled_kernel 24576 2104 $o/lookup_traffic_fp16.json "B=8, 48x64 queries, random coords +-8 px (tools/bench_lookup.py), fp16 tiled pyramid"
cat $o/bench_line.json
