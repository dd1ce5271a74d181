#!/usr/bin/env python3
"""Per-layer table of one gen_resnet call from a rocprofv3 --kernel-trace CSV of bench.py
(the 8 conv_mfma launches of a generator call repeat in order)."""
import csv
import glob
import sys

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof/*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(path)))
convs = [r for r in rows if "conv_mfma" in r["Kernel_Name"]]
names = ["b0.A 5x5 1->2", "b0.B 5x5 2->8 +s", "b1.A 5x5 8->128", "b1.B 5x5 128->128 +s", "b2.A 5x5 128->32",
         "b2.B 5x5 32->8 +s", "b3.A 5x5 8->2", "b3.B 5x5 2->1 +s"]
per_pass = {}
ncall = len(convs) // 8
# the first half of a volume's calls are pass 1 (input 64^2 upsampled), the second half pass 2
calls_per_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for c in range(ncall):
    ps = (c // calls_per_pass) % 2
    for k in range(8):
        r = convs[c * 8 + k]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        per_pass.setdefault((ps, k), []).append(d)
print("%-24s %-10s %10s %10s" % ("layer", "kernel", "pass1 us", "pass2 us"))
tot = [0, 0]
for k in range(8):
    t = convs[k]["Kernel_Name"].split("<")[1].split(">")[0]
    m = [sum(per_pass[(ps, k)]) / len(per_pass[(ps, k)]) for ps in (0, 1)]
    tot[0] += m[0]; tot[1] += m[1]
    print("%-24s %-10s %10.1f %10.1f" % (names[k], t, m[0], m[1]))
print("%-24s %-10s %10.1f %10.1f" % ("total", "", tot[0], tot[1]))
