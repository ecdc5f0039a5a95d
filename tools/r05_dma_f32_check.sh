#!/bin/bash
# Round 5: conv_dma.hip's fp32-input route (LDS-DMA + in-place conversion) - parity tests, then A/B against conv_patch.hip
set -u
out=$PWD/gpurun_out/r05
mkdir -p "$out"
export PYTHONUNBUFFERED=1
tag=${TAG:-f32}
timeout -k 10 1000 python -m pytest tests -q -m gpu --maxfail=40 > "$out/pytest_$tag.log" 2>&1
rc=$?
tail -25 "$out/pytest_$tag.log"
grep -E "^(FAILED|ERROR)" "$out/pytest_$tag.log" | head -40; [ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do FF_DMA_F32=$v python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_DMA_F32=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms; conv launches summed', d['roofline_conv']['sum_launch_ms'], 'ms; parity', d.get('secondary', {}).get('parity'))"; done | tee "$out/ab_dma_f32_$tag.txt"
for v in 1 0; do FF_DMA_F32=$v python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train FF_DMA_F32=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'loss', d['final_loss'])"; done | tee -a "$out/ab_dma_f32_$tag.txt"
python tools/conv_table.py > "$out/conv_table_$tag.txt" 2>&1; head -20 "$out/conv_table_$tag.txt"
