#!/bin/bash
out=gpurun_out/g16; mkdir -p $out
bash tools/collect_profiles_r04.sh > $out/collect.log 2>&1; echo "collect rc=$?" | tee -a $out/summary.txt
timeout 700 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 900 python -X faulthandler -m pytest tests -m gpu -q -n 1 --timeout 300 -rfE --tb=short -p no:cacheprovider > $out/guard_all.log 2>&1; echo "guard_all rc=$?" | tee -a $out/summary.txt
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$? wall ${SECONDS}s" | tee -a $out/summary.txt
python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/summary.txt
tail -3 $out/tests.log | cut -c1-200; tail -3 $out/guard_all.log | cut -c1-200; grep -c VIOLATION $out/guard_all.log; tail -2 $out/smoke.log | cut -c1-200; tail -12 $out/collect.log | cut -c1-200
