"""TensorFlow Saver-V2 ("tensor bundle") checkpoints without TensorFlow: ``<prefix>.index`` +
``<prefix>.data-00000-of-00001`` <-> {variable name: numpy array}.

The reference saves / restores its networks with ``tf.train.Saver`` (multipassGAN-out.py:367-386,
multipassGAN-4x.py:905-914); its published models are such files.  Format, restated from the TensorFlow
sources (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*, both Apache-2.0 and not vendored
here):

* ``.index`` is a LevelDB-style sorted string table: data blocks of prefix-compressed (key, value) entries
  with a restart array, each followed by a 1-byte compression tag and a masked crc32c; an index block mapping
  separator keys to block handles; a 48-byte footer (metaindex handle, index handle, padding, magic
  0xdb4775248b80fb57).  The bundle writer uses no block compression.
* key "" holds a BundleHeaderProto (num_shards = 1, little endian); every other key is a variable name with a
  BundleEntryProto value: dtype (1), shape (2), shard_id (3), offset (4), size (5), crc32c (6).
* ``.data-*`` is the concatenation of the raw little-endian tensor bytes.

STATUS: no TensorFlow-written file exists in this container to test against.  The reader is pinned by (a) the
published crc32c check values (RFC 3720 B.4) and LevelDB's mask, (b) index files assembled by a second, independent
builder in tests/test_host_tools.py (several data blocks, restart intervals 1 / 3 / 16 with and without prefix
sharing, corrupted block and tensor checksums, out-of-order keys, a snappy-tagged block), and (c) round trips through
the writer below.  float32 / float64 / int32 / int64 tensors, single shard, no slices.
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}


class CheckpointFormatError(Exception):
    pass


# ---------------------------------------------------------------- varints / protobuf
def _get_varint(buf, pos):
    result, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise CheckpointFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_message(buf):
    """-> list of (field number, wire type, value) of one protobuf message"""
    pos, fields = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointFormatError("unsupported protobuf wire type %d" % wt)
        fields.append((num, wt, v))
    return fields


def _parse_shape(buf):
    dims = []
    for num, _, v in _parse_message(buf):
        if num == 2:                                    # repeated Dim dim = 2 { int64 size = 1; }
            size = 0
            for n2, _, v2 in _parse_message(v):
                if n2 == 1:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(size)
    return tuple(dims)


def _field(num, wt, payload):
    return _put_varint((num << 3) | wt) + payload


def _entry_proto(dtype_id, shape, offset, size, crc):
    shp = b"".join(_field(2, 2, _put_varint(len(d)) + d) for d in (_field(1, 0, _put_varint(int(s))) for s in shape))
    msg = _field(1, 0, _put_varint(dtype_id)) + _field(2, 2, _put_varint(len(shp)) + shp)
    if offset:
        msg += _field(4, 0, _put_varint(offset))
    msg += _field(5, 0, _put_varint(size)) + _field(6, 5, struct.pack("<I", crc))
    return msg


# ---------------------------------------------------------------- crc32c (Castagnoli), masked as LevelDB does
_CRC_TABLE = []


def _crc32c(data):
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    c = 0xffffffff
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


# ---------------------------------------------------------------- sorted string table
def _read_block(data, offset, size, verify=True):
    """block contents at `offset`; the 5-byte trailer (compression tag, masked crc32c of contents + tag) is checked"""
    if offset + size + 5 > len(data):
        raise CheckpointFormatError("block handle (%d, %d) points past the end of the file" % (offset, size))
    tag = data[offset + size]
    if tag != 0:
        raise CheckpointFormatError("compressed index block (type %d): only uncompressed bundles are supported" % tag)
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if stored != _mask(_crc32c(data[offset:offset + size + 1])):
            raise CheckpointFormatError("block at offset %d fails its crc32c" % offset)
    return data[offset:offset + size]


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointFormatError("short block")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    if end < 0:
        raise CheckpointFormatError("restart array larger than its block")
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise CheckpointFormatError("corrupt block entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path, verify=True):
    """-> {key bytes: value bytes} of a sorted string table file (block checksums verified unless verify=False)"""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
        raise CheckpointFormatError("%s is not a tensor-bundle index (bad magic)" % path)
    foot = data[len(data) - 48:]
    _, p = _get_varint(foot, 0)
    _, p = _get_varint(foot, p)
    ioff, p = _get_varint(foot, p)
    isize, p = _get_varint(foot, p)
    table = {}
    last = None
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        for k, v in _block_entries(_read_block(data, boff, bsize, verify)):
            if last is not None and k <= last:
                raise CheckpointFormatError("keys out of order in the index (%r after %r)" % (k, last))
            table[k] = v
            last = k
    return table


def read_checkpoint(prefix, verify=True):
    """{variable name: numpy array} of the Saver-V2 checkpoint ``prefix`` (.index + .data-00000-of-00001); block and
    tensor crc32c values are verified unless verify=False"""
    table = read_index(prefix + ".index", verify)
    header = dict((n, v) for n, _, v in _parse_message(table.get(b"", b"")))
    if header.get(1, 1) != 1:
        raise CheckpointFormatError("%d shards: only single-shard checkpoints are supported" % header.get(1))
    if header.get(2, 0) != 0:
        raise CheckpointFormatError("big-endian bundle")
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        blob = f.read()
    out = {}
    for key, val in table.items():
        if key == b"":
            continue
        e = {"dtype": 0, "shape": (), "offset": 0, "size": 0, "slices": False, "crc": None}
        for num, _, v in _parse_message(val):
            if num == 1:
                e["dtype"] = v
            elif num == 2:
                e["shape"] = _parse_shape(v)
            elif num == 4:
                e["offset"] = v
            elif num == 5:
                e["size"] = v
            elif num == 6:
                e["crc"] = v
            elif num == 7:
                e["slices"] = True
        name = key.decode("utf-8")
        if e["slices"] or e["dtype"] not in DTYPES:
            continue                                     # partitioned variables / strings: not on this path
        dt = np.dtype(DTYPES[e["dtype"]])
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if n * dt.itemsize != e["size"] or e["offset"] + e["size"] > len(blob):
            raise CheckpointFormatError("entry %s: %d bytes for shape %s" % (name, e["size"], e["shape"]))
        if verify and e["crc"] is not None and e["crc"] != _mask(_crc32c(blob[e["offset"]:e["offset"] + e["size"]])):
            raise CheckpointFormatError("tensor %s fails its crc32c" % name)
        out[name] = np.frombuffer(blob, dtype=dt.newbyteorder("<"), count=n, offset=e["offset"]).reshape(e["shape"]).astype(dt)
    return out


def write_checkpoint(prefix, tensors, block_size=4096):
    """writes ``tensors`` ({name: array}) in the same format (one shard, uncompressed blocks, restart interval 16)"""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    blob, entries = bytearray(), []
    for nme in names:
        a = np.asarray(tensors[nme], order="C")                 # (np.ascontiguousarray would turn 0-d into 1-d)
        if a.dtype not in DTYPE_IDS:
            raise CheckpointFormatError("dtype %s of %s is not supported" % (a.dtype, nme))
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        entries.append((nme.encode("utf-8"), _entry_proto(DTYPE_IDS[a.dtype], a.shape, len(blob), len(raw), _mask(_crc32c(raw)))))
        blob += raw
    header = _field(1, 0, _put_varint(1)) + _field(3, 2, _put_varint(2) + _field(1, 0, _put_varint(1)))   # num_shards 1, version.producer 1
    items = [(b"", header)] + entries

    def build_block(kvs):
        out, restarts, last = bytearray(), [], b""
        for i, (k, v) in enumerate(kvs):
            shared = 0
            if i % 16 == 0:
                restarts.append(len(out))
            else:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            last = k
        for r in restarts:
            out += struct.pack("<I", r)
        out += struct.pack("<I", len(restarts))
        return bytes(out)

    out = bytearray()
    index, cur, cur_bytes = [], [], 0

    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        blk = build_block(cur)
        index.append((cur[-1][0], _put_varint(len(out)) + _put_varint(len(blk))))
        out.extend(blk + b"\x00" + struct.pack("<I", _mask(_crc32c(blk + b"\x00"))))
        cur, cur_bytes = [], 0

    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v)
        if cur_bytes >= block_size:
            flush()
    flush()
    meta = build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out.extend(meta + b"\x00" + struct.pack("<I", _mask(_crc32c(meta + b"\x00"))))
    idx = build_block(index)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out.extend(idx + b"\x00" + struct.pack("<I", _mask(_crc32c(idx + b"\x00"))))
    foot = meta_handle + idx_handle
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC))
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(blob))
