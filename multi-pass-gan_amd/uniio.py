"""mantaflow ``.uni`` grid files: same functions and header dictionary as the
reference's ``tools_wscale/uniio.py`` (readUni :81-88, writeUni :91-123).

File = gzip stream of: 4-byte magic (b"MNT2" or b"MNT3"), a 288-byte header,
then the raw C-order payload [dimZ, dimY, dimX, channels] of float32 (scalar:
elementType 1, 4 bytes/element; vec3: elementType 2, 12 bytes/element) or int32
(elementType 0).  Files are always written as MNT3.
"""
import gzip
import os
import shutil
import struct

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


def readUni(filename):
    """-> (header dict, ndarray [Z,Y,X,C])"""
    with gzip.open(filename, "rb") as stream:
        head = _parse_header(stream)
        return head, _parse_content(stream, head)


def writeUni(filename, header, content):
    """header: dict as returned by readUni (field order as in the file); content: array with
    dimX*dimY*dimZ (*3 for vec3) elements, converted to float32."""
    with gzip.open(filename, "wb") as stream:
        stream.write(b"MNT3")
        stream.write(struct.pack(_V4_FORMAT, *[header[k] for k in _V4_FIELDS]))
        content = np.asarray(content)
        if content.dtype != np.float32:
            content = content.astype(np.float32)
        n = header["dimX"] * header["dimY"] * header["dimZ"] * (3 if header["elementType"] == 2 else 1)
        stream.write(memoryview(np.ascontiguousarray(content).reshape(n)))


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
