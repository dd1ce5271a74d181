#!/usr/bin/env python3
"""Per-layer table of one gen_resnet call from a rocprofv3 --kernel-trace CSV of bench.py (the 7 conv
launches of a generator call repeat in order: resBlock 0 is one launch since round 3; bench.py
pipelines pass 1 of one volume with pass 2 of the previous one, so the mean is taken over the calls of both passes)."""
import csv
import glob
import sys

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(path)))
convs = [r for r in rows if "conv_mfma" in r["Kernel_Name"] or "conv_small" in r["Kernel_Name"]]
convs.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["b0 5x5 1->2->8 +s (pair)", "b1.A 5x5 8->128", "b1.B 5x5 128->128 +s", "b2.A 5x5 128->32",
         "b2.B 5x5 32->8 +s", "b3.A 5x5 8->2", "b3.B 5x5 2->1 +s"]
PLAN = [True] + [False] * 6
NL = len(names)
first = next(i for i, r in enumerate(convs) if "pair" in r["Kernel_Name"])
convs = convs[first:]
# generator calls = the leading run of launch groups in plan order (behind them: the roofline replays and the training block)
ncall = 0
while (ncall + 1) * NL <= len(convs) and ["pair" in convs[ncall * NL + k]["Kernel_Name"] for k in range(NL)] == \
        PLAN:
    ncall += 1
assert ncall >= 64, "only %d generator calls in plan order at the head of the trace" % ncall
per = {}
for c in range(ncall):
    for k in range(NL):
        r = convs[c * NL + k]
        per.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-24s %-34s %10s" % ("layer", "kernel", "mean us"))
tot = 0.0
for k in range(NL):
    kn = convs[k]["Kernel_Name"]
    short = kn.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    m = sum(per[k]) / len(per[k])
    tot += m
    print("%-24s %-34s %10.1f" % (names[k], short, m))
print("%-24s %-34s %10.1f" % ("total", "%d generator calls" % ncall, tot))
