#!/bin/bash
# Round-3 evidence run (on the GPU box, through gpurun): bench line, kernel trace and counter passes of the SAME command,
# training trace, lookup lab, per-layer conv table.  Everything lands in gpurun_out/r03/.
set -u
out=$PWD/gpurun_out/r03
mkdir -p "$out"
export PYTHONUNBUFFERED=1
echo "== default bench line"; python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"; tail -c 400 "$out/bench_line.json"
echo "== kernel trace of the bench command"
bash tools/prof_trace.sh r03/bench_b8 bench.py --no-secondary --no-cpu-baseline
echo "== counter passes of the bench command"
PASSES="1 2 4 6 7" bash tools/prof_pmc.sh r03/bench_pmc bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > "$out/bench_pmc.log" 2>&1
echo "== training step"
python bench.py --mode train --steps 6 --warmup 3 > "$out/train_line.json" 2> "$out/train_line.err"
bash tools/prof_trace.sh r03/train_b8 bench.py --mode train --steps 4 --warmup 2
echo "== A/B switches (training)"
for v in 0 1 0 1; do FF_PREPACK=$v FF_UNPACK_GROUP=$v python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_PREPACK=FF_UNPACK_GROUP=$v', d['value'], d['ms_per_step'])"; done | tee "$out/ab_train.txt"
for v in 1536 512 1536 512; do FF_WGRAD_BLOCKS=$v python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_WGRAD_BLOCKS=$v', d['value'], d['ms_per_step'])"; done | tee -a "$out/ab_train.txt"
echo "== weight-gradient table"
python tools/wgrad_table.py > "$out/wgrad_table.txt" 2>&1; head -4 "$out/wgrad_table.txt"
echo "== memory-only kernels of the lookup's size"
hipcc -O2 --offload-arch=gfx950 -o gpurun_out/hbm_gather tools/proto/hbm_gather.hip && timeout -k 10 120 ./gpurun_out/hbm_gather 16 > "$out/hbm_gather.log" 2>&1; grep "72.0 MB" "$out/hbm_gather.log"
echo "== lookup lab"
bash tools/lookup_lab.sh 8 48 64 0 8 50 > "$out/lookup_lab_b8_fp32.log" 2>&1; tail -8 "$out/lookup_lab_b8_fp32.log"
./gpurun_out/lookup_lab 8 48 64 1 8 50 > "$out/lookup_lab_b8_fp16.log" 2>&1
./gpurun_out/lookup_lab 4 68 120 1 8 50 > "$out/lookup_lab_c5_b4_fp16.log" 2>&1; tail -6 "$out/lookup_lab_c5_b4_fp16.log"
./gpurun_out/lookup_lab 16 48 64 0 8 50 > "$out/lookup_lab_b16_fp32.log" 2>&1
echo "== conv table"
python tools/conv_table.py > "$out/conv_table.txt" 2>&1; head -12 "$out/conv_table.txt"
echo "== norm kernels"
python tools/bench_norm.py > "$out/bench_norm.txt" 2>&1
ls "$out"
