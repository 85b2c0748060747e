#!/bin/bash
# after the barrier fix in conv_mfma_kernel: twins again, the asynchronous training twins, the two-rank gloo parity of PixelSNAIL
cd "$(dirname "$0")/../.."
f() { grep "^\[" | grep -v "^\[alone\] 0 of" | tail -12; }
echo "== twins (forward, 24 repeats)"; MODE=twins timeout 150 python tools/exp/conc_forward_selfcheck.py pixel_snail 24 2>&1 | f
echo "== asynchronous training twins"; timeout 150 python tools/exp/two_proc_nosync.py 2>&1 | grep "^c[01]"
echo "== bench.py --dp-parity, two ranks over gloo on one GPU, PixelSNAIL (three runs)"
export PG_FORCE_DEVICE=0 PG_DIST_BACKEND=gloo
for i in 1 2 3; do
timeout 200 python bench.py --gpus 2 --dp-parity --model pixel_snail --steps 4 --warmup 1 --batch 32 2>/tmp/x.err | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['dp_parity']; print('pixel_snail', p['ok'], 'vs rank0', p['max_abs_diff_vs_rank0'], 'vs 1-rank', p['max_abs_diff_vs_one_rank_run'])" || { echo "FAILED"; tail -2 /tmp/x.err | cut -c1-300; }
done
