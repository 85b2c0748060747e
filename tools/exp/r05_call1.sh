#!/bin/bash
# Round 5, first GPU call (gpurun --timeout 1100 -- 'bash tools/exp/r05_call1.sh'):
#   1. the whole GPU tier (with the new K-replays == K-eager-steps tests, the advised conv cases, --dp-parity over gloo)
#   2. same-box throughput: production library against the -DPG_BUFLOAD and max-ilp variants prepared in round 4;
#      the op + model tier with a variant ONLY if it gains >= 2 % on PixelSNAIL or GatedPixelCNN
#   3. counter passes of the headline's attention kernels and the weight-gradient kernels (profiles/r05_*)
#   4. the default bench line (the box's baseline for the round)
ulimit -c 0
R=$PWD
L=$R/pytorch-generative_amd/pytorch_generative_amd/lib
OUT=gpurun_out/c1; mkdir -p $OUT
echo "== 1. GPU tier"
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tier.log 2>&1
echo "rc=$? $(tail -1 $OUT/tier.log)"; grep -E "^(FAILED|ERROR)" $OUT/tier.log | head -20
echo "== 2. throughput, same box (images/s): prod | bufload | ilp"
declare -A best
for m in pixel_snail:1024 gated_pixel_cnn:512 pixel_cnn:1024 vd_vae:512; do
  M=${m%%:*}; B=${m##*:}
  line="$M"
  for v in prod bufload ilp; do
    so=$L/libpg_hip_$v.so; [ $v = prod ] && so=$L/libpg_hip.so
    [ -f $so ] || continue
    r=$(timeout 120 python tools/exp/bench_with_lib.py $so --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    line="$line  $v ${r#*: }"
    echo "$M $v ${r#*: }" >> $OUT/ab.txt
  done
  echo "$line"
done
for v in bufload ilp; do
  gain=$(python - <<PY
import collections
d = collections.defaultdict(dict)
for ln in open("$OUT/ab.txt"):
    p = ln.split()
    if len(p) == 3:
        d[p[0]][p[1]] = float(p[2])
g = [d[m]["$v"] / d[m]["prod"] for m in ("pixel_snail", "gated_pixel_cnn") if "$v" in d[m] and "prod" in d[m]]
print(1 if g and max(g) >= 1.02 else 0)
PY
)
  if [ "$gain" = "1" ]; then
    echo "== $v gains >= 2 %: op + model tier with it"
    PG_HIP_LIB=$L/libpg_hip_$v.so timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider > $OUT/tests_$v.log 2>&1
    echo "rc=$? $(tail -1 $OUT/tests_$v.log)"
  else
    echo "== $v: no gain >= 2 % on PixelSNAIL / GatedPixelCNN"
  fi
done
echo "== 3. counter passes"
bash tools/collect_profiles_r05.sh pmc 2>&1 | tail -30
echo "== 3b. kernel table of ImageGPT at the reference-default batch 64"
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/b64_stats -o p -- \
  python $R/bench.py --model image_gpt --steps 20 --warmup 3 --batch 64 --no-cpu-baseline > $R/$OUT/b64_stats.log 2>&1)
f=$(find $OUT/b64_stats -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/image_gpt_b64_kernel_stats.csv && head -14 $f | cut -c1-150
rm -rf $OUT/b64_stats
echo "== 4. default bench line"
cd $R
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$? $(cut -c1-400 $OUT/bench.json)"
