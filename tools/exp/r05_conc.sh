#!/bin/bash
# profiles/README.md round 5 item 16: which step of PixelSNAIL's forward changes when two processes share the GPU
cd "$(dirname "$0")/../.."
f() { grep "^\[" | grep -v "^\[alone\] 0 of" | tail -14; }
echo "== twins"; MODE=twins timeout 120 python tools/exp/conc_forward_selfcheck.py pixel_snail 16 2>&1 | f
echo "== shifted"; MODE=shifted timeout 120 python tools/exp/conc_forward_selfcheck.py pixel_snail 16 2>&1 | f
echo "== matmul"; MODE=matmul timeout 120 python tools/exp/conc_forward_selfcheck.py pixel_snail 16 2>&1 | f
