import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    for k in ("head_fwd", "head_bwd", "tail_fwd", "tail_bwd"):
        if k in r["Name"]:
            print(k, round(float(r["AverageNs"]) / 1e3, 1), end="  ")
print()
