#!/bin/bash
out=gpurun_out/g4; mkdir -p $out
timeout 200 python tests/guard/selftest.py > $out/selftest.log 2>&1; echo "selftest rc=$?" | tee -a $out/summary.txt
timeout 240 python tools/exp/b3_phase_prof.py 512 > $out/phase_p.log 2>&1; echo "phase_p rc=$?" | tee -a $out/summary.txt
PG_CONV_B3P=0 timeout 240 python tools/exp/b3_phase_prof.py 512 > $out/phase_old.log 2>&1; echo "phase_old rc=$?" | tee -a $out/summary.txt
PG_EXTRA_TESTS=1 timeout 700 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
PG_EXTRA_TESTS=1 PG_GUARD=1 PG_GUARD_POISON=0 AMD_SERIALIZE_KERNEL=3 timeout 600 python -X faulthandler -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 120 -rfE --tb=short -p no:cacheprovider > $out/guard_ops.log 2>&1; echo "guard_ops rc=$?" | tee -a $out/summary.txt
tail -12 $out/selftest.log | cut -c1-400; cat $out/phase_p.log | head -30; head -12 $out/phase_old.log;  tail -15 $out/tests.log | cut -c1-300; tail -30 $out/guard_ops.log | cut -c1-300
