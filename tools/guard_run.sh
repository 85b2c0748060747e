#!/bin/bash
# Memory-safety runs of the GPU test tier (through gpurun). Usage: tools/guard_run.sh <out-subdir> [repro] [guard] [guard-strict]
#   repro        the full tier in sequence, default allocator, full stderr kept
#   guard        PG_GUARD=1: every tensor between canary margins, contents poisoned (tests/guard/); one xdist worker so
#                that a GPU fault costs one test; PG_TRACE + AMD_SERIALIZE_KERNEL=3 name the launch a fault came from
#   guard-strict the same with 16-byte size rounding (every byte past numel() is canary)
out=gpurun_out/$1; shift
mkdir -p "$out"
export PG_EXTRA_TESTS=1
for mode in "$@"; do
  case $mode in
    repro)
      timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider > "$out/repro.log" 2>&1
      echo "repro rc=$?" >> "$out/summary.txt" ;;
    guard|guard-strict)
      align=512
      [ $mode = guard-strict ] && align=16
      mkdir -p "$out/$mode"
      # capture arena: never recycled, and the tier now captures every bench workload at full model size twice
      # (test_k_graph_replays_equal_k_eager_steps): 32 GB ran out in round 5 -> 128 GB of the 288
      PG_GUARD=1 PG_GUARD_ALIGN=$align PG_GUARD_ARENA_MB=${PG_GUARD_ARENA_MB:-131072} AMD_SERIALIZE_KERNEL=3 PG_TRACE="$out/$mode/trace" \
        timeout ${PG_GUARD_TIMEOUT:-1500} python -X faulthandler -m pytest tests -m gpu -q -n 1 --timeout 300 -rfE --tb=short -p no:cacheprovider ${PG_GUARD_PYTEST_ARGS} > "$out/$mode.log" 2>&1
      echo "$mode rc=$?" >> "$out/summary.txt"
      for f in "$out/$mode"/trace.*; do tail -n 40 "$f" > "$f.tail"; rm -f "$f"; done ;;
  esac
done
cat "$out/summary.txt"
for m in "$@"; do echo "== $m"; tail -n 25 "$out/$m.log"; done
