"""mantaflow ``.uni`` grid files: same functions and header dictionary as the
reference's ``tools_wscale/uniio.py`` (readUni :81-88, writeUni :91-123).

File = gzip stream of: 4-byte magic (b"MNT2" or b"MNT3"), a 288-byte header,
then the raw C-order payload [dimZ, dimY, dimX, channels] of float32 (scalar:
elementType 1, 4 bytes/element; vec3: elementType 2, 12 bytes/element) or int32
(elementType 0).  Files are always written as MNT3.  Large grids stream through a chunked, multi-threaded codec
(below) whose output every gzip reader, the reference's included, decodes to the same bytes.
"""
import concurrent.futures
import gzip
import io
import os
import shutil
import struct
import zlib

import numpy as np

_V4_FIELDS = ("dimX", "dimY", "dimZ", "gridType", "elementType", "bytesPerElement", "info", "dimT", "timestamp")
_V4_FORMAT = "iiiiii252siQ"      # MNT3, uniio.py:67
_V3_FORMAT = "iiiiii256sQ"       # MNT2, uniio.py:56
HEADER_BYTES = 288


class UniError(Exception):
    pass


def _parse_header(stream):
    magic = stream.read(4)
    raw = stream.read(HEADER_BYTES)
    if len(raw) != HEADER_BYTES:
        raise UniError("truncated .uni header")
    if magic == b"MNT3":
        head = dict(zip(_V4_FIELDS, struct.unpack(_V4_FORMAT, raw)))
    elif magic == b"MNT2":
        dx, dy, dz, gt, et, bpe, info, ts = struct.unpack(_V3_FORMAT, raw)
        # re-packed as a v4 header with dimT = 0 and the info string cut to 252 bytes (uniio.py:58-62)
        head = dict(zip(_V4_FIELDS, (dx, dy, dz, gt, et, bpe, info[0:252], 0, ts)))
    elif magic in (b"M4T2", b"M4T3"):
        raise UniError("4D .uni grids are not supported (uniio.py:69-71)")
    else:
        raise UniError("unknown .uni header %r" % (magic,))
    return head


def _parse_content(stream, head):
    et, bpe = head["elementType"], head["bytesPerElement"]
    if not ((bpe == 12 and et == 2) or (bpe == 4 and et in (0, 1))):
        raise UniError("unsupported element type %d with %d bytes per element" % (et, bpe))
    data = np.frombuffer(stream.read(), dtype="int32" if et == 0 else "float32")
    channels = 3 if et == 2 else 1
    dims = [head["dimZ"], head["dimY"], head["dimX"], channels]
    if head["dimT"] > 1:
        dims = [head["dimT"]] + dims
    return data.reshape(dims)


# ----------------------------------------------------------------------------------------------
# streaming codec.  A .uni file is ONE gzip stream for the reference (gzip.open(...).read()); RFC 1952 lets such a
# stream be a sequence of members, which every gzip reader -- the reference's included -- concatenates.  Volumes are
# therefore written as independently deflated chunks (one member each, compressed on a thread pool: zlib releases
# the GIL), and each member carries its own compressed size in a gzip "extra" subfield (the BGZF idea), so that this
# reader can find the members without inflating them and inflate them in parallel.  Files from other writers (one
# member, no subfield) are read as a plain stream.  A 512^3 volume is 537 MB: single-threaded level-9 deflate (the
# reference's gzip.open default) takes tens of seconds, longer than the three network passes that produce it.
# ----------------------------------------------------------------------------------------------
CHUNK_BYTES = 8 << 20
_SUBFIELD = b"MP"                 # extra-field subfield id: uint32 member size (header + deflate data + trailer)
_MEMBER_HEAD = 10 + 2 + 4 + 4     # fixed header, XLEN, subfield header, subfield payload


def _threads(threads):
    if threads is None:
        threads = int(os.environ.get("MPG_UNI_THREADS", "0")) or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    return max(1, int(threads))


def _deflate_member(buf, level):
    """one gzip member for `buf`, its total size recorded in the extra field"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(buf) + co.flush()
    total = _MEMBER_HEAD + len(body) + 8
    head = struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 255, 8) + _SUBFIELD + struct.pack("<HI", 4, total)
    return head + body + struct.pack("<II", zlib.crc32(buf) & 0xffffffff, len(buf) & 0xffffffff)


def _member_sizes(raw):
    """sizes of the members of a file written by writeUni, or None if it is not one"""
    sizes, at = [], 0
    n = len(raw)
    while at < n:
        if n - at < _MEMBER_HEAD or raw[at:at + 4] != b"\x1f\x8b\x08\x04" or raw[at + 12:at + 14] != _SUBFIELD:
            return None
        (size,) = struct.unpack_from("<I", raw, at + 16)
        if size < _MEMBER_HEAD + 8 or at + size > n:
            return None
        sizes.append(size)
        at += size
    return sizes


def _inflate_member(raw, at, size):
    data = zlib.decompress(raw[at + _MEMBER_HEAD:at + size - 8], -15)
    crc, isize = struct.unpack_from("<II", raw, at + size - 8)
    if (zlib.crc32(data) & 0xffffffff) != crc or (len(data) & 0xffffffff) != isize:
        raise UniError("corrupt gzip member at offset %d" % at)
    return data


def readUni(filename, threads=None):
    """-> (header dict, ndarray [Z,Y,X,C]).  Members written by writeUni are inflated in parallel."""
    with open(filename, "rb") as f:
        raw = f.read()
    sizes = _member_sizes(raw)
    if sizes is None:                       # a foreign writer (the reference, mantaflow): plain gzip stream
        with gzip.open(io.BytesIO(raw), "rb") as stream:
            head = _parse_header(stream)
            return head, _parse_content(stream, head)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
    nthreads = min(_threads(threads), len(sizes))
    if nthreads > 1:
        with concurrent.futures.ThreadPoolExecutor(nthreads) as pool:
            parts = list(pool.map(lambda a: _inflate_member(raw, a[0], a[1]), zip(offs, sizes)))
    else:
        parts = [_inflate_member(raw, o, z) for o, z in zip(offs, sizes)]
    stream = io.BytesIO(b"".join(parts))
    head = _parse_header(stream)
    return head, _parse_content(stream, head)


def writeUni(filename, header, content, level=None, threads=None, chunk_bytes=CHUNK_BYTES):
    """header: dict as returned by readUni (field order as in the file); content: array with dimX*dimY*dimZ (*3 for
    vec3) elements, converted to float32.  The decompressed bytes are those of the reference writer; the container is
    a multi-member gzip stream (see above).  level: deflate level (default MPG_UNI_LEVEL or 6; the reference's 9)."""
    if level is None:
        level = int(os.environ.get("MPG_UNI_LEVEL", "6"))
    content = np.asarray(content)
    if content.dtype != np.float32:
        content = content.astype(np.float32)
    n = header["dimX"] * header["dimY"] * header["dimZ"] * (3 if header["elementType"] == 2 else 1)
    payload = memoryview(np.ascontiguousarray(content).reshape(n)).cast("B")
    first = b"MNT3" + struct.pack(_V4_FORMAT, *[header[k] for k in _V4_FIELDS])
    cuts = list(range(0, len(payload), chunk_bytes)) or [0]
    pieces = [payload[c:c + chunk_bytes] for c in cuts]

    def member(i):
        return _deflate_member(first + bytes(pieces[0]) if i == 0 else pieces[i], level)

    nthreads = min(_threads(threads), len(pieces))
    with open(filename, "wb") as out:
        if nthreads > 1:
            with concurrent.futures.ThreadPoolExecutor(nthreads) as pool:
                for blob in pool.map(member, range(len(pieces))):       # in order; members stream out as they finish
                    out.write(blob)
        else:
            for i in range(len(pieces)):
                out.write(member(i))


def writeUniFromDevice(filename, header, volume, level=None, threads=None, chunk_bytes=CHUNK_BYTES):
    """writeUni for a float32 CUDA tensor: the volume is copied to pinned host memory chunk by chunk on a side stream
    while the previous chunks are being deflated, so neither the 537 MB device-to-host copy of a 512^3 volume nor its
    compression waits for the other."""
    import torch
    if level is None:
        level = int(os.environ.get("MPG_UNI_LEVEL", "6"))
    flat = volume.contiguous().reshape(-1)
    if flat.dtype != torch.float32:
        flat = flat.float()
    n = header["dimX"] * header["dimY"] * header["dimZ"] * (3 if header["elementType"] == 2 else 1)
    if flat.numel() != n:
        raise UniError("volume has %d elements, header describes %d" % (flat.numel(), n))
    per = max(1, chunk_bytes // 4)
    cuts = list(range(0, n, per)) or [0]
    host = torch.empty(n, dtype=torch.float32, pin_memory=flat.is_cuda)
    side = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None
    events = []
    if side is not None:
        side.wait_stream(torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(side):
            for c in cuts:
                host[c:c + per].copy_(flat[c:c + per], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
                events.append(ev)
    else:
        host.copy_(flat)
    view = memoryview(host.numpy()).cast("B")
    first = b"MNT3" + struct.pack(_V4_FORMAT, *[header[k] for k in _V4_FIELDS])

    def member(i):
        if events:
            events[i].synchronize()
        piece = view[cuts[i] * 4:(cuts[i] + per) * 4]
        return _deflate_member(first + bytes(piece) if i == 0 else piece, level)

    nthreads = min(_threads(threads), len(cuts))
    with open(filename, "wb") as out, concurrent.futures.ThreadPoolExecutor(nthreads) as pool:
        for blob in pool.map(member, range(len(cuts))):
            out.write(blob)


def make_header(dim_x, dim_y, dim_z, vec3=False, info=b"", timestamp=0, grid_type=1):
    """a fresh v4 header (the reference always copies one from an existing file, multipassGAN-out.py:629)"""
    return {
        "dimX": int(dim_x), "dimY": int(dim_y), "dimZ": int(dim_z), "gridType": int(grid_type),
        "elementType": 2 if vec3 else 1, "bytesPerElement": 12 if vec3 else 4,
        "info": bytes(info).ljust(252, b"\0")[:252], "dimT": 0, "timestamp": int(timestamp),
    }


def backupFile(name, test_path):
    """copy a source file into the run directory (uniio.py:126-130)"""
    shutil.copy(name if os.path.dirname(name) else "./" + name, test_path + os.path.basename(name))


# numpy array helpers (uniio.py:168-206)
def writeNumpySingle(filename, content):
    np.savez_compressed(filename, content)


def readNumpy(filename):
    return np.load(filename)
