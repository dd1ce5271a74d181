import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
nl = int(sys.argv[2])
plan = [l.strip()[5:] for l in open(sys.argv[3]) if l.startswith("PLAN")]
rows = [r for r in csv.DictReader(open(path)) if "conv_mfma" in r["Kernel_Name"]]
rows = rows[-nl:]
tot = 0
for r, p in zip(rows, plan):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%8.1f us  %-28s %s" % (d, r["Kernel_Name"].split("::")[-1].split("(")[0], p))
print("%8.1f us total" % tot)
