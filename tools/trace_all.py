"""per-launch table of the LAST generator call in a rocprofv3 --kernel-trace CSV of tools/trace_net.py: every kernel
between the last two occurrences of the call's first kernel, in start order, beside the plan lines of the log"""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
plan = [l.strip()[5:] for l in open(sys.argv[2]) if l.startswith("PLAN")]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
ncall = 4
# the call is periodic: find the period as the number of launches per call
n = len(rows)
per = None
for cand in range(4, n // 2):
    tail = [r["Kernel_Name"] for r in rows[-cand:]]
    prev = [r["Kernel_Name"] for r in rows[-2 * cand:-cand]]
    if tail == prev:
        per = cand
        break
rows = rows[-per:]
pi, tot = 0, 0.0
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    nm = r["Kernel_Name"].split("::")[-1].split("(")[0][:40]
    p = ""
    if "conv_mfma" in r["Kernel_Name"] or "conv_small" in r["Kernel_Name"]:
        p = plan[pi] if pi < len(plan) else ""
        pi += 1
    print("%8.1f us  %-40s %s" % (d, nm, p))
print("%8.1f us of kernels, %8.1f us wall for the call" % (tot, (int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
