import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), round(100 * float(r["TotalDurationNs"]) / tot, 1))
