#!/bin/bash
# Row-ring weight-gradient kernel: parity cases, then launch times with the ring on (production library) and off (ab library).
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -k "test_conv" -p no:cacheprovider > gpurun_out/r06_ring_tests.log 2>&1
tail -2 gpurun_out/r06_ring_tests.log
{
echo "== ring on (production library)"
python tools/exp/wgrad_ab.py "$@" 2>/dev/null
if [ -f pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so ]; then
echo "== ring off (ab library, PG_WGRAD_B3_RING=0)"
PG_HIP_LIB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so PG_WGRAD_B3_RING=0 python tools/exp/wgrad_ab.py "$@" 2>/dev/null
fi
} 2>&1 | tee gpurun_out/r06_wgrad_ring_ab.txt
