"""development probe: the matrix layers of the 4x generator as single F16F6 launches (8 slices of 256^2) with the K loop
or the conversion + stores switched off (mpg_conv_desc.reserved: 1 = no K loop, 2 = no output conversion / stores):
where does a launch spend its time?  Columns in microseconds; alternated A/B/A/B so clock drift shows."""
import sys
sys.path.insert(0, ".")
src = open("tools/probe_layers.py").read().split("layers = [")[0]
exec(src)
layers = [("b1.A 8->128", 8, 128, 5, None), ("b1.B 128->128+s8", 128, 128, 5, 8), ("b2.A 128->32", 128, 32, 5, None),
          ("b2.B 32->8+s128", 32, 8, 5, 128)]
print("%-20s %8s %8s %8s %8s | repeat" % ("layer F16F6", "full", "noK", "noStore", "neither"))
for name, cin, cout, k, ex in layers:
    t = [run(cin, cout, k, ex, d, 2, iters=30) for d in (0, 1, 2, 3, 0, 1, 2, 3)]
    print("%-20s " % name + " ".join("%8.1f" % v for v in t[:4]) + " | " + " ".join("%8.1f" % v for v in t[4:]), flush=True)
