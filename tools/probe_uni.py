"""f3: seconds per 512^3 volume (537 MB) through the .uni codec on this host: device -> pinned host -> chunked parallel
deflate -> file, and back; beside single-threaded gzip level 9 (what the reference's writeUni does)."""
import gzip, os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np, torch
import mpgan_amd
from mpgan_amd import uniio
from mpgan_amd.synthetic import synthetic_volume
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
low = synthetic_volume(64, 1, 0)[..., 0]
import scipy.ndimage
vol = np.maximum(scipy.ndimage.zoom(low, n / 64.0, order=1), 0).astype(np.float32)
vol[vol < 5e-4] = 0
dev = torch.as_tensor(vol).cuda()
h = uniio.make_header(n, n, n)
d = tempfile.mkdtemp()
p = os.path.join(d, "v.uni")
print("threads available:", len(os.sched_getaffinity(0)))
for lvl in (1, 6, 9):
    torch.cuda.synchronize(); t = time.time(); uniio.writeUniFromDevice(p, h, dev, level=lvl); tw = time.time() - t
    t = time.time(); hh, a = uniio.readUni(p); tr = time.time() - t
    assert np.array_equal(a[..., 0], vol)
    print("%d^3 level %d: write (from device) %.2f s, read %.2f s, %.1f MB" % (n, lvl, tw, tr, os.path.getsize(p) / 1e6), flush=True)
t = time.time(); host = dev.cpu().numpy(); t1 = time.time() - t
t = time.time()
with gzip.open(p, "wb") as f:
    f.write(b"MNT3" + b"\0" * 288); f.write(memoryview(host.reshape(-1)))
print("%d^3 reference path: D2H %.2f s + gzip.open level 9 single thread %.2f s, %.1f MB" % (n, t1, time.time() - t, os.path.getsize(p) / 1e6))
