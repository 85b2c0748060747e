#!/bin/bash
# Round 5: re-sweep of the ImageGPT launch parameters with the ab library (batch 1024, 30 timed steps each)
export PG_HIP_LIB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
run() { python bench.py --model image_gpt --batch ${B:-1024} --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | cut -d' ' -f2; }
echo "default $(run)"
for w in 8,8,8 4,8,8 8,8,4; do echo "PG_ATTN_WAVES=$w $(PG_ATTN_WAVES=$w run)"; done
for w in 4 6; do echo "PG_ATTN_BWD_WAVES=$w $(PG_ATTN_BWD_WAVES=$w run)"; done
for m in 1,1 2,2 2,4; do echo "PG_BLOCK_MINTILES=$m $(PG_BLOCK_MINTILES=$m run)"; done
for g in 2048,1024,2048,384 2048,1024,2048,640 2048,1280,2048,512 1024,1024,2048,512 3072,1024,2048,512 2048,1024,3072,512; do echo "PG_BLOCK_GRID=$g $(PG_BLOCK_GRID=$g run)"; done
echo "default again $(run)"
echo "-- batch 64"
B=64; echo "default $(B=64 run)"
for m in 1,1 1,2 2,2; do echo "PG_BLOCK_MINTILES=$m $(B=64 PG_BLOCK_MINTILES=$m run)"; done
for w in 4,8,4 8,8,8; do echo "PG_ATTN_WAVES=$w $(B=64 PG_ATTN_WAVES=$w run)"; done
