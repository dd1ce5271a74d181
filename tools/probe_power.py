"""development probe: board power and shader clock (rocm-smi, polled from a thread) while one layer runs back to back:
b1.B (5x5 128->128, 8 slices of 256^2) at F16F6 / F16X3 / F16X1, and the 8->128 and 128->32 layers at F16F6"""
import subprocess
import sys
import threading
import time

sys.path.insert(0, ".")
import torch
import mpgan_amd  # noqa: F401
from mpgan_amd import ops

dev = "cuda:0"


def poll(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "-P", "-c", "--csv"], capture_output=True, text=True, timeout=5).stdout
            rows.append(out.strip().splitlines()[-1])
        except Exception as e:                      # noqa: BLE001
            rows.append("error %s" % e)
        time.sleep(0.3)


def run(name, cin, cout, prec, secs=4.0):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((8, 256, 256, cin), device=dev, generator=g).relu_()
    w = torch.randn((5, 5, cin, cout), device=dev, generator=g)
    segs = [ops.Segment(x, ops.pack_conv_weights(w, wscale=0.05, prec=prec))]
    call = lambda: ops.conv2d_fused(segs, (256, 256), act="relu", want_f32=False, want_g8=True)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, rows))
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            call()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    print("%-22s %.1f us per launch;  rocm-smi samples (last 6):" % (name, e0.elapsed_time(e1) * 1e3 / n))
    for r in rows[-6:]:
        print("     ", r)


hdr = subprocess.run(["rocm-smi", "-d", "0", "-P", "-c", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print("idle:", hdr[0] if hdr else "", "|", hdr[-1] if hdr else "")
run("b1.B F16F6", 128, 128, 2)
run("b1.B F16X3", 128, 128, 3)
run("b1.B F16X1", 128, 128, 1)
run("b2.A F16F6 (128->32)", 128, 32, 2)
run("b1.A F16X3 (8->128)", 8, 128, 3)
