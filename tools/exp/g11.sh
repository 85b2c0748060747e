#!/bin/bash
out=gpurun_out/g11; mkdir -p $out
b() { tag=$1; shift; env "$@" timeout 200 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$M $B $tag', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" | tee -a $out/bench.txt; }
M=pixel_snail; B=1024
b w8 PG_X=1; b old PG_CONV_B3P=0; b w8 PG_X=1; b old PG_CONV_B3P=0
B=128; b w8 PG_X=1; b old PG_CONV_B3P=0
M=beta_vae; B=1024; b w8 PG_X=1; b old PG_CONV_B3P=0
timeout 700 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 900 python -X faulthandler -m pytest tests -m gpu -q -n 1 --timeout 300 -rfE --tb=short -p no:cacheprovider > $out/guard_all.log 2>&1; echo "guard_all rc=$?" | tee -a $out/summary.txt
cat $out/bench.txt; tail -3 $out/tests.log | cut -c1-200; tail -4 $out/guard_all.log | cut -c1-200; grep -n "VIOLATION\|crashed" $out/guard_all.log | head
