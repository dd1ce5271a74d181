"""development probe: the four matrix layers of gen_resnet at MPG_PREC_F16F8, isolated, repeated"""
import sys
sys.path.insert(0, ".")
src = open("tools/probe_layers.py").read().split("layers = [")[0]
exec(src)
layers = [("b1.A 8->128", 8, 128, 5, None), ("b1.B 128->128+s8", 128, 128, 5, 8), ("b2.A 128->32", 128, 32, 5, None),
          ("b2.B 32->8+s128", 32, 8, 5, 128)]
for name, cin, cout, k, ex in layers:
    t = [run(cin, cout, k, ex, 0, 2, iters=40) for _ in range(3)]
    print("%-20s " % name + " ".join("%8.1f" % v for v in t), flush=True)
