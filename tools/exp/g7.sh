#!/bin/bash
out=gpurun_out/g7; mkdir -p $out
timeout 240 python tools/exp/b3_phase_prof.py 512 > $out/phase_w8.log 2>&1; echo "phase_w8 rc=$?" | tee -a $out/summary.txt
PG_CONV_B3P_WAVES=4 timeout 240 python tools/exp/b3_phase_prof.py 512 > $out/phase_w4.log 2>&1; echo "phase_w4 rc=$?" | tee -a $out/summary.txt
timeout 700 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 900 python -X faulthandler -m pytest tests -m gpu -q -n 1 --timeout 300 -rfE --tb=short -p no:cacheprovider > $out/guard_all.log 2>&1; echo "guard_all rc=$?" | tee -a $out/summary.txt
head -24 $out/phase_w8.log; head -10 $out/phase_w4.log; tail -6 $out/tests.log | cut -c1-250; tail -12 $out/guard_all.log | cut -c1-250
