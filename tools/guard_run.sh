#!/bin/bash
# Memory-safety runs of the GPU test tier (through gpurun). Usage: tools/guard_run.sh <out-subdir> [repro] [guard-end] [guard-start] [guard-strict]
#   repro        the full tier with the formerly opt-in cases in sequence, default allocator, full stderr kept
#   guard-end    PG_GUARD=1, tensors flush against the END guard page (overruns), one xdist worker so a fault
#                costs one test; PG_TRACE + AMD_SERIALIZE_KERNEL=3 name the launch
#   guard-start  the same with tensors flush against the START guard page (underruns)
#   guard-strict END guard with 16-byte rounding (every byte past numel() faults)
out=gpurun_out/$1; shift
mkdir -p "$out"
export PG_EXTRA_TESTS=1
for mode in "$@"; do
  case $mode in
    repro)
      timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider > "$out/repro.log" 2>&1
      echo "repro rc=$?" >> "$out/summary.txt" ;;
    guard-end|guard-start|guard-strict)
      side=end; align=512
      [ $mode = guard-start ] && side=start
      [ $mode = guard-strict ] && align=16
      mkdir -p "$out/$mode"
      PG_GUARD=1 PG_GUARD_SIDE=$side PG_GUARD_ALIGN=$align AMD_SERIALIZE_KERNEL=3 PG_TRACE="$out/$mode/trace" \
        timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -n 1 -p no:cacheprovider ${PG_GUARD_PYTEST_ARGS} > "$out/$mode.log" 2>&1
      echo "$mode rc=$?" >> "$out/summary.txt"
      for f in "$out/$mode"/trace.*; do tail -n 40 "$f" > "$f.tail"; rm -f "$f"; done ;;
  esac
done
cat "$out/summary.txt"
for m in "$@"; do echo "== $m"; tail -n 25 "$out/$m.log"; done
