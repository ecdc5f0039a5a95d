#!/bin/bash
# Round 5: the recorded update loop as one autograd node - gradient tests, then the training line with and without it.
set -u
out=$PWD/gpurun_out/r05
mkdir -p "$out"
export PYTHONUNBUFFERED=1
tag=${TAG:-loop}
timeout -k 10 900 python -m pytest tests/test_hip_backward.py tests/test_param_gate.py tests/test_ddp_training.py tests/test_shim_dropin.py tests/test_full_size.py -x -q -m gpu > "$out/pytest_train_$tag.log" 2>&1
rc=$?
tail -15 "$out/pytest_train_$tag.log"
[ $rc -ne 0 ] && exit $rc
python tools/host_issue_time_train.py > "$out/train_host_$tag.txt" 2>&1; cat "$out/train_host_$tag.txt"
for v in 1 0 1; do FF_TRAIN_LOOP=$v python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FF_TRAIN_LOOP=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'loss', d['final_loss'])"; done | tee "$out/ab_train_loop_$tag.txt"
