#!/bin/bash
# Runs ON THE GPU BOX: the kernel sequence of ONE replayed step of a workload (rocprofv3 --kernel-trace), to see where stray
# launches (copyBuffer, ATen fills) sit.   bash tools/exp/step_sequence.sh <model> <batch>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/seq_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o p -- python $R/bench.py --model $1 --steps 4 --warmup 2 --batch $2 --no-cpu-baseline > $OUT/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/tr/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "lr_decay_kernel" in n or "adam_kernel" in n]
ends = [i for i, n in enumerate(names) if "adam_kernel" in n]
a, b = ends[-2], ends[-1]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
with open("$OUT/sequence.txt", "w") as out:
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        out.write(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {n[:100]}\n")
        prev_end = e
print(open("$OUT/sequence.txt").read()[:200])
PY
rm -rf $OUT/tr
