"""development probe: fused-conv launches of the matrix layers with the K loop or the stores disabled
(mpg_conv_desc.reserved: 1 skip K loop, 2 skip stores), F16F6"""
import sys
sys.path.insert(0, ".")
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("pl", "tools/probe_layers.py")
src = open("tools/probe_layers.py").read().split("layers = [")[0]
exec(src)
layers = [("b1.A 8->128", 8, 128, 5, None), ("b1.B 128->128+s8", 128, 128, 5, 8), ("b2.A 128->32", 128, 32, 5, None),
          ("b2.B 32->8+s128", 32, 8, 5, 128)]
layers = [("b1.A 8->128", 8, 128, 5, None), ("b1.B 128->128+s8", 128, 128, 5, 8)]
print("%-20s  alternating full / stagger" % "layer F16F6")
for name, cin, cout, k, ex in layers:
    t = [run(cin, cout, k, ex, d, 2, iters=40) for d in (0, 4, 0, 4, 0, 4)]
    print("%-20s " % name + " ".join("%8.1f" % v for v in t), flush=True)
