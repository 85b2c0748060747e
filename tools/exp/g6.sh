#!/bin/bash
out=gpurun_out/g6; mkdir -p $out
timeout 200 python tests/guard/selftest.py > $out/selftest.log 2>&1; echo "selftest rc=$?" | tee -a $out/summary.txt
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 PG_TRACE=$out/trace_vae timeout 300 python -X faulthandler -m pytest tests/test_gpu_reference_suite.py -m gpu -q -s -x --timeout 200 -k "integration_reproduce" -p no:cacheprovider > $out/guard_ref.log 2>&1; echo "guard_ref rc=$?" | tee -a $out/summary.txt
for f in $out/trace_vae.*; do tail -n 30 "$f" > "$f.tail"; rm -f "$f"; done
export PG_EXTRA_TESTS=1
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 timeout 900 python -X faulthandler -m pytest tests -m gpu -q -n 1 --timeout 300 -rfE --tb=short -p no:cacheprovider > $out/guard_all.log 2>&1; echo "guard_all rc=$?" | tee -a $out/summary.txt
tail -8 $out/selftest.log | cut -c1-300; grep -n "Memory access\|pg_guard\]\|HSA\|fault" $out/guard_ref.log | head; tail -5 $out/guard_ref.log | cut -c1-200; for f in $out/trace_vae.*.tail; do echo "== $f"; tail -4 $f | cut -c1-400; done; tail -40 $out/guard_all.log | cut -c1-250
