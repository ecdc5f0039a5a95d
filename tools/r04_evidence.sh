#!/bin/bash
# Round-4 evidence run (on the GPU box, through gpurun, once at the end of the round; PART=1 and PART=2 split it
# into two calls of under 20 minutes): the default bench line, kernel
# trace and counter passes of the SAME command, one update iteration out of the trace, per-layer tables, in-kernel phase
# stamps of the lab build, training line.  Everything lands in gpurun_out/r04/.
set -u
out=$PWD/gpurun_out/r04
mkdir -p "$out"
export PYTHONUNBUFFERED=1
part=${PART:-12}
if [[ $part == *1* ]]; then
echo "== bench line (the command of the driver's BENCH record)"; python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_line.json" 2> "$out/bench_line.err"; tail -c 300 "$out/bench_line.json"; echo
echo "== A/B: split-pair activations in the update block (same box, 20 steps each)"
for v in 0 1 0 1; do FF_SPLIT_ACT=$v python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_SPLIT_ACT=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms; conv launches summed', d['roofline_conv']['sum_launch_ms'], 'ms; lookup frac', d['roofline']['frac'])"; done | tee "$out/ab_split_act.txt"
echo "== kernel trace of the bench command"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/tr" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > "$out/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/tr" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_b8_kernel_stats.csv"
f=$(find "$out/tr" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_iter.py "$f" > "$out/iter_trace.txt" 2>&1
[ -n "$f" ] && FULL=1 FWD=4 python tools/trace_encoder.py "$f" > "$out/encoder_trace.txt" 2>&1
rm -rf "$out/tr"; tail -4 "$out/iter_trace.txt"
echo "== counter passes of the bench command (conv_dma kernels, lookup)"
PASSES="1 2 3 4 6 8" bash tools/prof_pmc.sh r04/bench_pmc bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > "$out/bench_pmc.log" 2>&1
python tools/pmc_summary.py "$out/bench_pmc" conv_dma > "$out/conv_dma_pmc_summary.txt" 2>&1
python tools/pmc_summary.py "$out/bench_pmc" gru_pass > "$out/gru_pass_pmc_summary.txt" 2>&1
python tools/pmc_summary.py "$out/bench_pmc" fusion_pair > "$out/fusion_pair_pmc_summary.txt" 2>&1
python tools/pmc_summary.py "$out/bench_pmc" lookup_dma > "$out/lookup_pmc_summary.txt" 2>&1
python tools/make_traffic_json.py "$out/bench_pmc" lookup_dma_kernel 24576 2904 "$out/lookup_traffic.json" "the lookup launches of bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline itself (8 pairs 384x512, 12 iterations), tools/prof_pmc.sh passes 1 2 3 4 6 8 + a kernel-trace pass"
fi
if [[ $part == *2* ]]; then
echo "== per-layer tables"
python tools/conv_table.py > "$out/conv_table.txt" 2>&1; head -14 "$out/conv_table.txt"
TILES=8,8,4 python tools/bench_dma_conv.py 8 > "$out/dma_layers_b8.txt" 2>&1; cat "$out/dma_layers_b8.txt"
TILES=8,8 python tools/bench_dma_conv.py 32 > "$out/dma_layers_b32.txt" 2>&1; tail -3 "$out/dma_layers_b32.txt"
echo "== A/B: the fused GRU pass, the all-channels blocks (same box, 20 steps each)"
for v in 0 1 0 1; do FF_GRU_PASS=$v python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_GRU_PASS=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"; done | tee "$out/ab_gru_pass.txt"
for v in 0 1 0 1; do FF_FUSION_PAIR=$v python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_FUSION_PAIR=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"; done | tee "$out/ab_fusion_pair.txt"
for v in 0 1 0 1; do FF_DMA_ALLCH=$v python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_DMA_ALLCH=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"; done | tee "$out/ab_allch.txt"
python tools/loop_time.py 8 > "$out/loop_time.txt" 2>&1; cat "$out/loop_time.txt"
echo "== in-kernel stamps (lab build)"
if [ -f focusflow_official_amd/lib/libfocusflow_lab.so ]; then
  FF_LAB_LIB=libfocusflow_lab.so FF_GRU_PASS=0 python tools/dma_stamps.py 8 > "$out/dma_stamps_b8.txt" 2>&1; cat "$out/dma_stamps_b8.txt"
  FF_LAB_LIB=libfocusflow_lab.so python tools/gru_pass_stamps.py 8 > "$out/gru_pass_stamps.txt" 2>&1; cat "$out/gru_pass_stamps.txt"
fi
echo "== training step"
python bench.py --mode train --steps 6 --warmup 3 > "$out/train_line.json" 2> "$out/train_line.err"; tail -c 200 "$out/train_line.json"; echo
fi
ls "$out"
