#!/bin/bash
out=gpurun_out/g5; mkdir -p $out
timeout 200 python tests/guard/selftest.py > $out/selftest.log 2>&1; echo "selftest rc=$?" | tee -a $out/summary.txt
timeout 120 python tools/exp/run_coexec.py > $out/coexec.log 2>&1; echo "coexec rc=$?" | tee -a $out/summary.txt
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 700 python -X faulthandler -m pytest tests/test_gpu_ops.py tests/test_gpu_reference_suite.py -m gpu -q --timeout 200 -rfE --tb=short -p no:cacheprovider > $out/guard_ops.log 2>&1; echo "guard_ops rc=$?" | tee -a $out/summary.txt
tail -12 $out/selftest.log | cut -c1-400; cat $out/coexec.log; tail -50 $out/guard_ops.log | cut -c1-300
