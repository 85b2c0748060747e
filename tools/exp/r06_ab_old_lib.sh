L=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib
for spec in pixel_cnn:1024 pixel_snail:1024 vd_vae:512 gated_pixel_cnn:512; do
  M=${spec%%:*}; B=${spec##*:}
  for rep in 1 2; do
    n=$(python bench.py --model $M --batch $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
    o=$(python tools/exp/bench_with_lib.py $L/libpg_hip_old.so --model $M --batch $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
    echo "$M new $n old $o"
  done
done
