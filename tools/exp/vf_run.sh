mkdir -p gpurun_out/vf
timeout 280 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/vf/tier.log 2>&1; echo "tier rc=$?"; tail -2 gpurun_out/vf/tier.log
timeout 200 python bench.py > gpurun_out/vf/bench_new.json 2> gpurun_out/vf/bench_new.err; echo "bench rc=$?"
P=pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_prev.so
for m in pixel_snail:1024 image_gpt:1024; do
  M=${m%%:*}; B=${m##*:}
  timeout 90 python tools/exp/bench_with_lib.py $P --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/vf/prev_$M.json 2> gpurun_out/vf/prev_$M.err
  echo "prev $M: $(grep -o '"value": [0-9.]*' gpurun_out/vf/prev_$M.json | head -1)"
done
grep -o '"value": [0-9.]*' gpurun_out/vf/bench_new.json | head -1
