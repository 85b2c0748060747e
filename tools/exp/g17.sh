#!/bin/bash
out=gpurun_out/g17; mkdir -p $out
timeout 400 python -X faulthandler -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
b() { tag=$1; shift; env "$@" timeout 200 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$M $B $tag', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" | tee -a $out/bench.txt; }
M=vd_vae; B=512; b big PG_X=1; b plan PG_CONV_B3_BIGTILE=0
M=beta_vae; B=1024; b big PG_X=1; b plan PG_CONV_B3_BIGTILE=0
tail -3 $out/tests.log | cut -c1-200; cat $out/bench.txt
