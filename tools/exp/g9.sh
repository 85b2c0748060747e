#!/bin/bash
out=gpurun_out/g9; mkdir -p $out
timeout 120 python tools/exp/run_coexec.py > $out/coexec.log 2>&1; echo "coexec rc=$?" | tee -a $out/summary.txt
timeout 240 python tools/exp/b3_phase_prof.py 512 > $out/phase_w8.log 2>&1; echo "phase_w8 rc=$?" | tee -a $out/summary.txt
timeout 400 python -X faulthandler -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 300 python -X faulthandler -m pytest tests/test_gpu_reference_suite.py -m gpu -q -s -x --timeout 200 -k "vq_vae or vae-vae" -p no:cacheprovider > $out/guard_vq.log 2>&1; echo "guard_vq rc=$?" | tee -a $out/summary.txt
cat $out/coexec.log; head -24 $out/phase_w8.log; tail -5 $out/tests.log | cut -c1-250; grep -n "pg_guard\]" $out/guard_vq.log | head -5; tail -3 $out/guard_vq.log | cut -c1-200
