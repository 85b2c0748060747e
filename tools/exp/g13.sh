#!/bin/bash
out=gpurun_out/g13; mkdir -p $out
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","n_gpus")})
print("b64", d["imagegpt_b64"]["images_per_s"])
print("snail", d["pixel_snail"]["images_per_s"], d["pixel_snail"]["reference_default_batch_128"]["images_per_s"], {k:d["pixel_snail"]["roofline"][k] for k in ("achieved","peak","frac","launch_ms","traffic")})
for k,v in d["other_configs"].items():
    if isinstance(v,dict): print(k, round(v["images_per_s"],1), v.get("frac_of_fp32_compute_ceiling"), (v.get("dominant_kernel") or {}).get("kernel"))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","launch_ms","traffic")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["oracle_over_reference_step_time"])
PY
