#!/bin/bash
# First GPU call of the next round (through gpurun from the repo root, ~4 GPU-minutes): the experiment variants that were
# prepared after round 4's GPU budget was spent (profiles/README.md round 4, item 14).
#   PG_VARIANT=bufload python pytorch-generative_amd/build.py          # in the build container, BEFORE the call
#   PG_VARIANT=ilp PG_EXTRA_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp" PG_ALLOW_SPILLS=1 python pytorch-generative_amd/build.py
#   gpurun --timeout 600 -- 'bash tools/exp/next_round.sh'
ulimit -c 0
L=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib
OUT=gpurun_out/next_round; mkdir -p $OUT
for v in bufload ilp; do
  so=$L/libpg_hip_$v.so
  [ -f $so ] || { echo "== $v: $so not built, skipped"; continue; }
  echo "== $v: parity of the op tier and the model tier with this library"
  PG_HIP_LIB=$so timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider > $OUT/tests_$v.log 2>&1
  echo "rc=$? $(tail -1 $OUT/tests_$v.log)"
done
echo "== graph replay against eager launches, all workloads (production library)"
timeout 240 python tools/exp/graph_vs_eager_all.py 2>&1 | tail -9
echo "== throughput, same box: production library, then each variant (images/s)"
for m in pixel_snail:1024 gated_pixel_cnn:512 pixel_cnn:1024 vd_vae:512; do
  M=${m%%:*}; B=${m##*:}
  line="$M"
  for v in prod bufload ilp; do
    so=$L/libpg_hip_$v.so; [ $v = prod ] && so=$L/libpg_hip.so
    [ -f $so ] || continue
    r=$(timeout 90 python tools/exp/bench_with_lib.py $so --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    line="$line  $v ${r#*: }"
  done
  echo "$line"
done
