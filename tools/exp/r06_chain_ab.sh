#!/bin/bash
# Runs ON THE GPU BOX: ImageGPT with the model-level merged weight-gradient reduction (default) against one reduction per block
for B in 64 1024; do for rep in 1 2; do
  n=$(python bench.py --model image_gpt --batch $B --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  o=$(PG_BLOCK_CHAIN=0 python bench.py --model image_gpt --batch $B --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "image_gpt batch $B: chain $n  per-block $o"
done; done
