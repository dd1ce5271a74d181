"""development probe: times the F16F6 matrix layers (single launches, 8 slices of 256^2) with each library under
tools/variants/ (tools/build_variants.sh), one child process per library, round-robin twice so drift shows."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, ".")
src = open("tools/probe_layers.py").read().split("layers = [")[0]
exec(src)
layers = [("b1.A", 8, 128, 5, None), ("b1.B", 128, 128, 5, 8), ("b2.A", 128, 32, 5, None), ("b2.B", 32, 8, 5, 128)]
print(" ".join("%s %.1f" % (n, run(ci, co, k, ex, 0, 2, iters=40)) for n, ci, co, k, ex in layers), flush=True)
'''
libs = sorted(f for f in os.listdir(os.path.join(ROOT, "tools", "variants")) if f.endswith(".so"))
if len(sys.argv) > 1:
    libs = [l for l in libs if any(a in l for a in sys.argv[1:])]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, MPGAN_LIB_OVERRIDE=os.path.join(ROOT, "tools", "variants", lib))
        r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True)
        print("%-28s %s" % (lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
