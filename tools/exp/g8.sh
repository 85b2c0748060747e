#!/bin/bash
out=gpurun_out/g8; mkdir -p $out
PG_GUARD=1 AMD_SERIALIZE_KERNEL=3 PG_TRACE=$out/trace timeout 300 python -X faulthandler -m pytest tests/test_gpu_reference_suite.py -m gpu -q -s -x --timeout 200 -k "vq_vae" -p no:cacheprovider > $out/guard_vq.log 2>&1; echo "guard_vq rc=$?" | tee -a $out/summary.txt
for f in $out/trace.*; do tail -n 12 "$f" > "$f.tail"; rm -f "$f"; done
grep -n "Memory access\|pg_guard\]\|HSA\|fault\|VIOLATION" $out/guard_vq.log | head -20; for f in $out/trace.*.tail; do echo "== $f"; cat $f | cut -c1-300; done
