mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "incremental" 2>&1 | tail -15
python - <<'PY'
import sys, time, torch
sys.path.insert(0, "pytorch-generative_amd")
import pytorch_generative_amd as pg
m = pg.models.ImageGPT(in_channels=1, out_channels=1, in_size=28).to("cuda")
with torch.no_grad(): m(torch.zeros(2,1,28,28,device="cuda"))
for n in (64, 512):
    for inc in (True, False):
        if not inc and n > 64: continue
        torch.cuda.synchronize(); t=time.time(); s = m.sample(n_samples=n, incremental=inc); torch.cuda.synchronize()
        print(f"sample n={n} incremental={inc}: {time.time()-t:.2f} s")
PY
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
