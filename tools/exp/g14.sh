#!/bin/bash
out=gpurun_out/g14; mkdir -p $out
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
b() { tag=$1; shift; env "$@" timeout 200 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$M $B $tag', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" | tee -a $out/bench.txt; }
M=pixel_snail; B=1024
b fused PG_X=1; b twok PG_ATTN_FUSED_BWD_K4=0; b fused PG_X=1; b twok PG_ATTN_FUSED_BWD_K4=0
B=128; b fused PG_X=1; b twok PG_ATTN_FUSED_BWD_K4=0
python - <<'PY' | tee -a $out/bench.txt
import sys, torch
sys.path[:0] = [".", "pytorch-generative_amd"]
import bench
r = bench.attention_kernel_roofline(1024, torch.device("cuda:0"), 1, 4, 32, 32, True)
print({k: (round(v["launch_ms"], 4), round(v["tflops"], 1)) for k, v in r.items()})
PY
tail -4 $out/tests.log | cut -c1-250; cat $out/bench.txt
