"""Host-side drop-in modules (uniio, paramhelpers, FluidDataLoader) against fixtures produced
by the reference's own modules (tests/golden/make_golden.py)."""
import gzip
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "tools_golden.npz"))


def _header(uniio, n, vec3):
    h = uniio.make_header(n, n, n, vec3=vec3, info=b"golden", timestamp=1234567, grid_type=4 if vec3 else 1)
    return h


def test_uni_write_is_byte_identical(mpg, tmp_path):
    from mpgan_amd import uniio
    for kind, vec3 in (("scalar", False), ("vec3", True)):
        p = str(tmp_path / (kind + ".uni"))
        arr = GOLD["uni_%s_in" % kind]
        uniio.writeUni(p, _header(uniio, 6, vec3), arr)
        with gzip.open(p, "rb") as f:
            raw = np.frombuffer(f.read(), dtype=np.uint8)
        assert np.array_equal(raw, GOLD["uni_%s_bytes" % kind])      # magic + 288-byte header + payload
        head, back = uniio.readUni(p)
        assert np.array_equal(back, arr) and back.shape == arr.shape
        assert head["dimX"] == 6 and head["elementType"] == (2 if vec3 else 1) and head["timestamp"] == 1234567


def test_uni_reads_old_mnt2_header(mpg, tmp_path):
    from mpgan_amd import uniio
    p = str(tmp_path / "old.uni")
    with gzip.open(p, "wb") as f:
        f.write(GOLD["uni_mnt2_bytes"].tobytes())
    head, arr = uniio.readUni(p)
    assert np.array_equal(arr, GOLD["uni_scalar_in"])
    assert head["dimT"] == int(GOLD["uni_mnt2_dimT"]) and len(head["info"]) == int(GOLD["uni_mnt2_info_len"])
    # float64 content is converted, wrong magic is refused
    q = str(tmp_path / "f64.uni")
    uniio.writeUni(q, _header(uniio, 6, False), GOLD["uni_scalar_in"].astype(np.float64))
    assert np.array_equal(uniio.readUni(q)[1], GOLD["uni_scalar_in"])
    with gzip.open(str(tmp_path / "bad.uni"), "wb") as f:
        f.write(b"XXXX" + bytes(288))
    with pytest.raises(uniio.UniError):
        uniio.readUni(str(tmp_path / "bad.uni"))


@pytest.fixture()
def sim_dir(mpg, tmp_path):
    from mpgan_amd import uniio
    sim = tmp_path / "sim_1000"
    sim.mkdir()
    d_low, v_low, d_hi = GOLD["fdl_d_low"], GOLD["fdl_v_low"], GOLD["fdl_d_hi"]
    for f in range(d_low.shape[0]):
        uniio.writeUni(str(sim / ("density_low_%04d.uni" % f)), _header(uniio, 8, False), d_low[f])
        uniio.writeUni(str(sim / ("velocity_low_%04d.uni" % f)), _header(uniio, 8, True), v_low[f])
        uniio.writeUni(str(sim / ("density_high_%04d.uni" % f)), _header(uniio, 16, False), d_hi[f])
    return str(tmp_path) + "/"


def test_fluiddataloader_output_mode(mpg, sim_dir):
    from mpgan_amd.fluiddataloader import FluidDataLoader
    fl = FluidDataLoader(print_info=0, base_path=sim_dir, base_path_y=sim_dir, numpy_seed=42,
                         filename="density_low_%04d.uni", filename_index_min=0, oldNamingScheme=False, filename_y=None,
                         filename_index_max=3, indices=[1000], data_fraction=1.0,
                         multi_file_list=["density", "velocity"], multi_file_list_y=["density"])
    x, y, names = fl.get()
    assert y is None and x.dtype == np.float32
    assert np.array_equal(x, GOLD["fdl_out_x"])
    assert [os.path.basename(n) for n in names] == list(GOLD["fdl_out_names"])


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_fluiddataloader_slice_mode(mpg, sim_dir, axis):
    from mpgan_amd.fluiddataloader import FluidDataLoader
    fl = FluidDataLoader(print_info=0, base_path=sim_dir, base_path_y=sim_dir, numpy_seed=42, conv_slices=True,
                         conv_axis=axis, select_random=0.5, density_threshold=0.45,
                         axis_scaling_y=[0.5, 1, 1, 1] if axis == 0 else [1, 1, 1, 1],
                         axis_scaling=[1, 1, 1, 1] if axis == 0 else [2, 1, 1, 1],
                         filename="density_low_%04d.uni", oldNamingScheme=False, filename_y="density_high_%04d.uni",
                         filename_index_max=3, filename_index_min=0, indices=[1000], data_fraction=1.0,
                         multi_file_list=["density", "velocity"] * 3, multi_file_idxOff=[0, 0, 1, 1, 2, 2],
                         multi_file_list_y=["density"] * 3, multi_file_idxOff_y=[0, 1, 2])
    x, y, _ = fl.get()
    gx, gy = GOLD["fdl_slices_x_axis%d" % axis], GOLD["fdl_slices_y_axis%d" % axis]
    assert x.shape == gx.shape and y.shape == gy.shape
    assert np.array_equal(x, gx) and np.array_equal(y, gy)


def test_fluiddataloader_fraction_and_errors(mpg, sim_dir):
    from mpgan_amd.fluiddataloader import FluidDataLoader, FluidDataLoaderError
    fl = FluidDataLoader(print_info=0, base_path=sim_dir, base_path_y=sim_dir, numpy_seed=1,
                         filename="density_low_%04d.uni", filename_index_min=0, filename_index_max=4, indices=[1000],
                         data_fraction=0.5)
    x, _, names = fl.get()
    assert np.array_equal(x, GOLD["fdl_fraction_x"])
    assert [os.path.basename(n) for n in names] == list(GOLD["fdl_fraction_names"])
    with pytest.raises(FluidDataLoaderError):
        FluidDataLoader(print_info=0, base_path=sim_dir, filename="a_%04d.uni", wildcard="x", indices=[1000])
    with pytest.raises(FluidDataLoaderError):
        FluidDataLoader(print_info=0, base_path=sim_dir, filename="density_low_%04d.uni", indices=[1000],
                        filename_index_max=2, multi_file_list=["density", "velocity"], multi_file_idxOff=[0])


def test_paramhelpers(mpg, tmp_path, capsys):
    from mpgan_amd import paramhelpers as ph
    argv = ["prog", "upRes", "8", "TileSize", "64", "bogus", "1"]
    del ph.paramUsed[:]
    assert ph.getParam("upres", 4, argv) == "8"          # case-insensitive, strings returned
    assert ph.getParam("tileSize", 16, argv) == "64"
    assert ph.getParam("simSize", 64, argv) == 64        # default passes through untouched
    with pytest.raises(SystemExit):
        ph.checkUnusedParams(argv)
    assert "bogus" in capsys.readouterr().out
    p, no = ph.getNextGenericPath("out_0001-0002", 0, str(tmp_path) + "/")
    p2, no2 = ph.getNextGenericPath("out_0001-0002", 0, str(tmp_path) + "/")
    assert (no, no2) == (0, 1) and os.path.isdir(p) and p2.endswith("out_0001-0002_0001/")
    ph.writeParams(str(tmp_path / "params.json"))
    assert ph.readParams(str(tmp_path / "params.json"))["upres"] == "8"


def test_tf_saver_v2_bundle_round_trip(tmp_path):
    """tf_checkpoint: the restated tensor-bundle format (sorted string table index + raw data) round trips,
    several index blocks, prefix-compressed keys, scalars; checkpoint.load falls back to it.  No TensorFlow-written
    file is available to pin the reader against (parity unpinned)."""
    from mpgan_amd import checkpoint, tf_checkpoint as T
    rng = np.random.default_rng(0)
    d = {"generator/g_cA0/weight": rng.standard_normal((5, 5, 1, 2)).astype(np.float32),
         "generator/g_cA0/bias": np.full(2, 0.1, np.float32), "global_step": np.array(7, np.int64),
         "generator/g_cB1/weight": rng.standard_normal((5, 5, 16, 24)).astype(np.float32),
         "beta1_power": np.array(0.5, np.float32), "d64": rng.standard_normal(3)}
    for i in range(300):
        d["generator/genBlock%d/g_cA_first/weight/ExponentialMovingAverage" % i] = rng.standard_normal((3, i % 5 + 1)).astype(np.float32)
    prefix = str(tmp_path / "test_0001" / "model_0002.ckpt")
    T.write_checkpoint(prefix, d, block_size=1024)
    r = T.read_checkpoint(prefix)
    assert sorted(r) == sorted(d)
    for k in d:
        assert r[k].dtype == d[k].dtype and r[k].shape == d[k].shape and np.array_equal(r[k], d[k]), k
    # index structure: footer magic, more than one data block
    raw = open(prefix + ".index", "rb").read()
    assert int.from_bytes(raw[-8:], "little") == T.MAGIC
    assert T._mask(T._crc32c(b"123456789")) == (((0xe3069283 >> 15) | (0xe3069283 << 17)) + 0xa282ead8) & 0xffffffff
    loaded = checkpoint.load(prefix)                   # no .npz next to it: the TF reader is used, float32 only
    assert "global_step" not in loaded and np.array_equal(loaded["generator/g_cA0/weight"], d["generator/g_cA0/weight"])
    with pytest.raises(T.CheckpointFormatError):
        bad = tmp_path / "bad.ckpt.index"
        bad.write_bytes(b"\\0" * 64)
        T.read_index(str(bad))


def test_uni_streaming_codec(mpg, tmp_path):
    """f3: chunked multi-member gzip writer / parallel reader: the decompressed stream is the reference writer's bytes
    (fixture), any gzip reader concatenates the members, foreign single-member files still read, corruption is detected"""
    import gzip
    from mpgan_amd import uniio
    dens = GOLD["uni_scalar_in"]
    want = GOLD["uni_scalar_bytes"].tobytes()
    n = dens.shape[0]
    head = {"dimX": n, "dimY": n, "dimZ": n, "gridType": 1, "elementType": 1, "bytesPerElement": 4,
            "info": b"golden".ljust(252, b"\0"), "dimT": 0, "timestamp": 1234567}
    p = str(tmp_path / "m.uni")
    uniio.writeUni(p, head, dens, chunk_bytes=256, threads=3)            # 4 members
    raw = open(p, "rb").read()
    assert len(uniio._member_sizes(raw)) == 4
    with gzip.open(p, "rb") as f:
        assert f.read() == want                                          # what the reference's reader sees
    h, a = uniio.readUni(p, threads=2)
    assert np.array_equal(a, dens) and h["timestamp"] == 1234567
    # a foreign (single member) file
    q = str(tmp_path / "f.uni")
    with gzip.open(q, "wb") as f:
        f.write(want)
    h2, a2 = uniio.readUni(q)
    assert np.array_equal(a2, dens) and uniio._member_sizes(open(q, "rb").read()) is None
    # a flipped payload bit is caught by the member's crc
    bad = bytearray(raw)
    bad[len(bad) // 2] ^= 0x10
    open(p, "wb").write(bytes(bad))
    with pytest.raises((uniio.UniError, Exception)):
        uniio.readUni(p)
    # vec3 + larger, default settings
    v = np.random.default_rng(3).standard_normal((20, 20, 20, 3)).astype(np.float32)
    uniio.writeUni(p, uniio.make_header(20, 20, 20, vec3=True), v, chunk_bytes=10000)
    assert np.array_equal(uniio.readUni(p)[1], v)


# ---------------------------------------------------------------------------------------------
# f1: TF Saver-V2 reader against hand-assembled tables (second, independent builder) and known check values
# ---------------------------------------------------------------------------------------------
def _vi(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _table(blocks, restart_interval, share_prefixes, tf_ck, corrupt_block=None, tag=0):
    """LevelDB table: `blocks` = list of lists of (key, value); returns file bytes.  Written independently of
    tf_checkpoint.write_checkpoint: other restart intervals, optional absence of prefix sharing."""
    import struct
    out = bytearray()

    def block(kvs, interval, share):
        b, restarts, last = bytearray(), [], b""
        for i, (k, v) in enumerate(kvs):
            sh = 0
            if i % interval == 0:
                restarts.append(len(b))
            elif share:
                while sh < min(len(k), len(last)) and k[sh] == last[sh]:
                    sh += 1
            b += _vi(sh) + _vi(len(k) - sh) + _vi(len(v)) + k[sh:] + v
            last = k
        if not restarts:
            restarts = [0]
        for r in restarts:
            b += struct.pack("<I", r)
        return bytes(b + struct.pack("<I", len(restarts)))

    def emit(blk, t=0, bad=False):
        off = len(out)
        crc = tf_ck._mask(tf_ck._crc32c(blk + bytes([t])))
        out.extend(blk + bytes([t]) + struct.pack("<I", crc ^ (1 if bad else 0)))
        return _vi(off) + _vi(len(blk))

    handles = []
    for i, kvs in enumerate(blocks):
        handles.append((kvs[-1][0], emit(block(kvs, restart_interval, share_prefixes), tag if i == 0 else 0, corrupt_block == i)))
    meta = emit(block([], 1, False))
    idx = emit(block(handles, 1, False))
    foot = meta + idx
    out.extend(foot + b"\0" * (40 - len(foot)) + struct.pack("<Q", tf_ck.MAGIC))
    return bytes(out)


def test_tf_checkpoint_reader_on_hand_built_tables(mpg, tmp_path):
    import struct
    from mpgan_amd import tf_checkpoint as ck
    # crc32c check values of RFC 3720 B.4 and the LevelDB mask (Mask(crc32c("foo")) is not crc32c("foo"), and unmasking restores it)
    assert ck._crc32c(b"123456789") == 0xe3069283
    assert ck._crc32c(bytes(32)) == 0x8a9136aa and ck._crc32c(b"\xff" * 32) == 0x62a8ab43
    assert ck._crc32c(bytes(range(32))) == 0x46dd794e
    m = ck._mask(ck._crc32c(b"foo"))
    assert m != ck._crc32c(b"foo") and ((((m - 0xa282ead8) & 0xffffffff) >> 17) | (((m - 0xa282ead8) & 0xffffffff) << 15)) & 0xffffffff == ck._crc32c(b"foo")
    rng = np.random.default_rng(1)
    tensors = {"generator/genBlock%d/g_c%s_%s/%s" % (b, ab, nm, wb): rng.standard_normal(shp).astype(np.float32)
               for b in (2, 4, 8) for ab in "AB" for nm in ("first", "second") for wb, shp in (("weight", (3, 3, 2, 4)), ("bias", (4,)))}
    tensors["global_step"] = np.array(7, dtype=np.int64)
    names = sorted(tensors, key=lambda s: s.encode())
    blob, entries = bytearray(), []
    for nme in names:
        raw = tensors[nme].tobytes()
        entries.append((nme.encode(), ck._entry_proto(ck.DTYPE_IDS[tensors[nme].dtype], tensors[nme].shape, len(blob), len(raw),
                                                      ck._mask(ck._crc32c(raw)))))
        blob += raw
    header = b"\x08\x01"                                  # BundleHeaderProto { num_shards: 1 }
    items = [(b"", header)] + entries
    prefix = str(tmp_path / "model_0001.ckpt")
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(blob))
    for interval, share, nblocks in ((1, False, 1), (3, True, 4), (16, True, 2), (5, True, len(items))):
        per = -(-len(items) // nblocks)
        blocks = [items[i:i + per] for i in range(0, len(items), per)]
        open(prefix + ".index", "wb").write(_table(blocks, interval, share, ck))
        got = ck.read_checkpoint(prefix)
        assert sorted(got) == sorted(tensors)
        for k in tensors:
            assert got[k].dtype == tensors[k].dtype and np.array_equal(got[k], tensors[k]), (interval, k)
    blocks = [items[:9], items[9:]]
    # a data block whose checksum is wrong
    open(prefix + ".index", "wb").write(_table(blocks, 4, True, ck, corrupt_block=1))
    with pytest.raises(ck.CheckpointFormatError, match="crc32c"):
        ck.read_checkpoint(prefix)
    assert sorted(ck.read_checkpoint(prefix, verify=False)) == sorted(tensors)
    # a snappy-compressed block is refused, not mis-read
    open(prefix + ".index", "wb").write(_table(blocks, 4, True, ck, tag=1))
    with pytest.raises(ck.CheckpointFormatError, match="compressed"):
        ck.read_checkpoint(prefix)
    # keys out of order
    open(prefix + ".index", "wb").write(_table([items[9:], items[:9]], 4, True, ck))
    with pytest.raises(ck.CheckpointFormatError, match="order"):
        ck.read_checkpoint(prefix)
    # a flipped bit in a tensor
    open(prefix + ".index", "wb").write(_table(blocks, 4, True, ck))
    bad = bytearray(blob)
    bad[100] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(bad))
    with pytest.raises(ck.CheckpointFormatError, match="tensor"):
        ck.read_checkpoint(prefix)
    # wrong magic
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[-1] ^= 0xff
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ck.CheckpointFormatError, match="magic"):
        ck.read_checkpoint(prefix)
