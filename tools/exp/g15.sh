#!/bin/bash
out=gpurun_out/g15; mkdir -p $out
echo "== 8 waves" > $out/wgrad.txt; timeout 200 python tools/exp/wgrad_ab.py 2>&1 | grep -v amdgpu >> $out/wgrad.txt
echo "== 4 waves" >> $out/wgrad.txt; PG_WGRAD_B3_WAVES=4 timeout 200 python tools/exp/wgrad_ab.py 2>&1 | grep -v amdgpu >> $out/wgrad.txt
timeout 400 python -X faulthandler -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv" -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
b() { tag=$1; shift; env "$@" timeout 200 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$M $B $tag', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" | tee -a $out/bench.txt; }
M=pixel_snail; B=1024; b w8 PG_X=1; b w4 PG_WGRAD_B3_WAVES=4
M=gated_pixel_cnn; B=512; b w8 PG_X=1; b w4 PG_WGRAD_B3_WAVES=4
M=pixel_cnn; B=1024; b w8 PG_X=1; b w4 PG_WGRAD_B3_WAVES=4
cat $out/wgrad.txt; tail -3 $out/tests.log | cut -c1-200; cat $out/bench.txt
