"""Host-side plumbing: Kaldi table I/O wire format, nnet.config, blueprint loading."""

import io
import os
import struct
import sys

import numpy as np
import pytest

from libs.support import kaldi_io
import libs.support.utils as utils


def test_write_vec_flt_is_byte_exact():
    buf = io.BytesIO()
    v = np.arange(5, dtype=np.float32) * 0.5
    kaldi_io.write_vec_flt(buf, v, key="utt1")
    assert buf.getvalue() == b"utt1 \0BFV \x04" + struct.pack("<I", 5) + v.tobytes()
    buf.seek(0)
    assert kaldi_io.read_key(buf) == "utt1"
    assert np.array_equal(kaldi_io.read_vec_flt(buf), v)
    assert kaldi_io.read_key(buf) is None


def test_matrix_roundtrip_float_and_double(tmp_path):
    r = np.random.RandomState(0)
    mats = {"a": r.randn(7, 30).astype(np.float32), "b": r.randn(200, 80).astype(np.float32), "c64": r.randn(3, 4)}
    ark = tmp_path / "feats.ark"
    with open(ark, "wb") as f:
        for k, m in mats.items():
            kaldi_io.write_mat(f, m, key=k)
    got = dict(kaldi_io.read_mat_ark(str(ark)))
    assert list(got) == list(mats)
    for k in mats:
        assert got[k].dtype == mats[k].dtype and np.array_equal(got[k], mats[k])
    # header layout: key SP \0B FM \4 rows \4 cols
    raw = open(ark, "rb").read()
    assert raw.startswith(b"a \0BFM \x04" + struct.pack("<I", 7) + b"\x04" + struct.pack("<I", 30))
    # 'ark:' prefix and chunk reads
    assert np.array_equal(next(iter(kaldi_io.read_mat_ark("ark:" + str(ark))))[1], mats["a"])


def test_scp_offsets_and_chunk(tmp_path):
    r = np.random.RandomState(1)
    m = r.randn(50, 6).astype(np.float32)
    ark = tmp_path / "x.ark"
    with open(ark, "wb") as f:
        f.write(b"k1 ")
        off = f.tell()
        kaldi_io.write_mat(f, m)
    scp = tmp_path / "x.scp"
    scp.write_text("k1 %s:%d\n" % (ark, off))
    (key, got), = list(kaldi_io.read_mat_scp(str(scp)))
    assert key == "k1" and np.array_equal(got, m)
    assert np.array_equal(kaldi_io.read_mat("%s:%d" % (ark, off), chunk=[10, 19]), m[10:20])


def test_text_matrix_and_vector():
    txt = io.BytesIO(b" [\n  1 2 3\n  4 5 6 ]\n")
    assert np.array_equal(kaldi_io.read_mat(txt), np.array([[1, 2, 3], [4, 5, 6]], dtype=np.float32))
    assert np.array_equal(kaldi_io.read_vec_flt(io.BytesIO(b" [ 1.5 2 -3 ]\n")), np.array([1.5, 2, -3]))


def test_compressed_matrix_decodes_like_kaldi():
    """'CM ' = global header + per-column uint16 percentiles + column-major uint8 (kaldi_io.py:527-569)."""
    rows, cols = 5, 3
    gmin, grange = -2.0, 8.0
    pct = np.array([[0, 16384, 49152, 65535]] * cols, dtype=np.uint16)
    data = np.array([[0, 64, 128, 192, 255]] * cols, dtype=np.uint8)
    blob = b"\0BCM " + struct.pack("<ffii", gmin, grange, rows, cols) + pct.tobytes() + data.tobytes()
    mat = kaldi_io.read_mat(io.BytesIO(blob))
    p = pct[0].astype(np.float64) * grange / 65535.0 + gmin
    want = np.array([p[0], p[1], p[1] + (p[2] - p[1]) * 64 / 128.0, p[2], p[3]])
    assert mat.shape == (rows, cols)
    assert np.allclose(mat[:, 0], want, atol=1e-5)


def test_pipe_rspecifier_and_wspecifier(tmp_path):
    m = np.random.RandomState(2).randn(4, 5).astype(np.float32)
    ark = tmp_path / "p.ark"
    with open(ark, "wb") as f:
        kaldi_io.write_mat(f, m, key="u")
    with kaldi_io.open_or_fd("ark:cat %s |" % ark, "rb") as r:
        assert kaldi_io.read_key(r) == "u"
        assert np.array_equal(kaldi_io.read_mat(r), m)
    out = tmp_path / "out.ark"
    w = kaldi_io.open_or_fd("ark:| cat > %s" % out, "wb")
    kaldi_io.write_vec_flt(w, m[0], key="u")
    w.close()
    import time
    for _ in range(50):
        if out.exists() and out.stat().st_size > 0:
            break
        time.sleep(0.05)
    assert dict(kaldi_io.read_vec_flt_ark(str(out)))["u"].tolist() == m[0].tolist()


def test_batched_reader_packs_ragged_groups(tmp_path):
    r = np.random.RandomState(3)
    lens = [100, 200, 50, 300, 20]
    ark = tmp_path / "b.ark"
    with open(ark, "wb") as f:
        for i, n in enumerate(lens):
            kaldi_io.write_mat(f, r.randn(n, 8).astype(np.float32), key="u%d" % i)
    groups = list(kaldi_io.read_mat_ark_batched(str(ark), max_frames=350, max_utts=10))
    assert [g[0] for g in groups] == [["u0", "u1", "u2"], ["u3", "u4"]]
    keys, feats, offs = groups[0]
    assert feats.shape == (350, 8) and offs.tolist() == [0, 100, 300, 350] and offs.dtype == np.int32


def test_vec_ark_scp_writer(tmp_path):
    items = [("a", np.arange(3, dtype=np.float32)), ("b", np.arange(4, dtype=np.float32) + 1)]
    kaldi_io.write_vec_flt_ark_scp(str(tmp_path / "x.ark"), str(tmp_path / "x.scp"), items)
    got = dict(kaldi_io.read_vec_flt_scp(str(tmp_path / "x.scp")))
    assert np.array_equal(got["a"], items[0][1]) and np.array_equal(got["b"], items[1][1])
    got2 = dict(kaldi_io.read_vec_flt_auto("scp:" + str(tmp_path / "x.scp")))
    assert list(got2) == ["a", "b"]


def test_nnet_config_roundtrip_and_reference_format(tmp_path):
    cfg = tmp_path / "nnet.config"
    creation = "Xvector(30,10,training=False,extracted_embedding='near')"
    utils.write_nnet_config("model/xvector.py", creation, str(cfg))
    assert utils.read_nnet_config(str(cfg)) == ("model/xvector.py", creation)
    # what pandas DataFrame.to_csv(header=None, sep=';') writes in the reference (utils.py:189-193)
    cfg.write_text("model_blueprint;subtools/pytorch/model/xvector.py\nmodel_creation;Xvector(23,1211,training=False)\n")
    assert utils.read_nnet_config(str(cfg)) == ("subtools/pytorch/model/xvector.py", "Xvector(23,1211,training=False)")


def test_assign_params_dict_semantics():
    d = {"a": 1, "b": {"c": 2.0, "d": True}, "e": None}
    out = utils.assign_params_dict(d, {"a": 5, "b": {"c": 3}, "zzz": 1, "e": "x"})
    assert out == {"a": 5, "b": {"c": 3.0, "d": True}, "e": "x"}
    assert utils.assign_params_dict(d, {"zzz": 1}, support_unknow=True)["zzz"] == 1
    with pytest.raises(ValueError):
        utils.assign_params_dict(d, {"a": "str"})


def test_blueprints_build_with_reference_state_dict_keys():
    import helpers
    for name in ("xvector_c1", "ecapa_c3", "ecapa_c512_fc1_far"):
        g, shapes = helpers.load_golden(name)
        model = helpers.build_model(str(g["blueprint"]), str(g["creation"]))
        mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        assert mine == shapes, name                      # same keys, same order, same shapes as the reference model
        assert model.get_model_creation().startswith(str(g["creation"]).split("(")[0])


def test_training_side_is_absent_not_faked():
    import helpers
    with pytest.raises(NotImplementedError):
        helpers.build_model("xvector.py", "Xvector(30,10,training=True)")
    model = helpers.build_model("xvector.py", "Xvector(30,10,training=False)")
    import torch
    with pytest.raises(NotImplementedError):
        model.tdnn1(torch.zeros(1, 30, 50))
    with pytest.raises(RuntimeError):                     # CPU model: no fallback
        model.extract_embedding(np.zeros((50, 30), dtype=np.float32))


def test_packed_ark_reader_and_batch_vector_writer():
    """The bulk feed of the extraction script: block-parsed headers, payloads copied / read straight into a caller-owned
    packed buffer; float64 entries, entries longer than the buffer, group limits, streams without readinto; the batch
    vector writer emits exactly write_vec_flt's bytes."""
    import io
    from libs.support import kaldi_io as K
    rng = np.random.RandomState(0)
    mats = [rng.randn(t, 7).astype(np.float32) for t in (3, 1, 50, 200, 17, 999, 5)]
    bio = io.BytesIO()
    for i, m in enumerate(mats):
        K.write_mat(bio, m.astype(np.float64) if i == 2 else m, key="k%d" % i)
    data = bio.getvalue()

    class ReadOnly(object):
        def __init__(self, b):
            self.b = io.BytesIO(b)

        def read(self, n):
            return self.b.read(n)

    for block in (16, 1000, 1 << 20):
        for make in (io.BytesIO, ReadOnly):
            rd = K.PackedArkReader(make(data), block=block)
            assert rd.peek_dim() == 7
            buf, got, sizes = np.empty((256, 7), np.float32), [], []
            while True:
                keys, offs, n = rd.read_group(buf, max_utts=3)
                if not keys:
                    break
                sizes.append(len(keys))
                if isinstance(n, np.ndarray):                        # 999 frames do not fit the 256-frame buffer
                    got.append((keys[0], n.copy()))
                else:
                    got.extend((k, buf[offs[j]:offs[j + 1]].copy()) for j, k in enumerate(keys))
            assert [k for k, _ in got] == ["k%d" % i for i in range(7)] and max(sizes) <= 3
            for (k, g), m in zip(got, mats):
                assert np.array_equal(g, m), k
    assert K.PackedArkReader(io.BytesIO(b"")).peek_dim() is None
    with pytest.raises(K.BadInputFormat):
        rd = K.PackedArkReader(io.BytesIO(data[:-5]))
        rd.peek_dim()
        while rd.read_group(np.empty((2048, 7), np.float32))[0]:
            pass
    v = rng.randn(4, 5).astype(np.float32)
    blob = K.vec_flt_ark_bytes(["a", "bb", "c", "d"], v)
    ref = io.BytesIO()
    for i, k in enumerate(["a", "bb", "c", "d"]):
        K.write_vec_flt(ref, v[i], key=k)
    assert blob == ref.getvalue()


def _cm_entry(key, mat_u8, pct, gmin, grange):
    """One 'CM ' ark entry from explicit quantiser tables (what Kaldi's copy-feats --compress=true stores)."""
    cols, rows = mat_u8.shape
    return (key + " ").encode() + b"\0BCM " + struct.pack("<ffii", gmin, grange, rows, cols) + pct.astype(np.uint16).tobytes() + mat_u8.astype(np.uint8).tobytes()


def test_packed_ark_reader_generic_entries_after_peek_and_deferred():
    """compressed ('CM ', Kaldi's default for stored features) and text entries go through the generic
    decoder; peek_dim() decodes the first one early and an entry that does not fit the batch is deferred to the next
    read_group() - both leave a decoded ndarray in the pending slot, which must not be compared with a string."""
    from libs.support import kaldi_io as K
    rng = np.random.RandomState(5)
    cols = 6
    pct = np.sort(rng.randint(0, 65536, size=(cols, 4)), axis=1)
    q = [rng.randint(0, 256, size=(cols, rows)) for rows in (40, 30, 50, 3)]
    blob = b"".join(_cm_entry("c%d" % i, m, pct, -3.0, 9.0) for i, m in enumerate(q))
    blob += b"t0  [\n  " + b"\n  ".join(b" ".join(b"%g" % v for v in row) for row in np.arange(24).reshape(4, 6)) + b" ]\n"
    want = [K.read_mat(io.BytesIO(_cm_entry("", m, pct, -3.0, 9.0)[1:])) for m in q] + [np.arange(24, dtype=np.float32).reshape(4, 6)]
    for block in (16, 1 << 13):
        rd = K.PackedArkReader(io.BytesIO(blob), block=block)
        assert rd.peek_dim() == cols and rd.peek_dim() == cols          # peeking twice is harmless
        buf, got = np.empty((64, cols), np.float32), []
        while True:
            keys, offs, n = rd.read_group(buf)                          # 40 + 30 > 64: 'c1' is decoded, then deferred
            if not keys:
                break
            got.extend((k, buf[offs[j]:offs[j + 1]].copy()) for j, k in enumerate(keys))
        assert [k for k, _ in got] == ["c0", "c1", "c2", "c3", "t0"]
        for (k, g), w in zip(got, want):
            assert g.shape == w.shape and np.array_equal(g, w), k
    # a text ark whose first entry is peeked, with a buffer smaller than that entry
    rd = K.PackedArkReader(io.BytesIO(blob[blob.index(b"t0 "):]))
    assert rd.peek_dim() == cols
    keys, offs, big = rd.read_group(np.empty((2, cols), np.float32))
    assert keys == ["t0"] and isinstance(big, np.ndarray) and np.array_equal(big, want[-1])


def test_pipe_writer_waits_for_the_child_and_reports_its_status(tmp_path):
    """`ark:| copy-vector ark:- ark,scp:...` - close() returns only when the child has finished writing, and a
    failing child raises in the caller (the reference waits in a non-daemon thread, kaldi_io.py:76-113)."""
    from libs.support import kaldi_io as K
    out = tmp_path / "slow.ark"
    w = K.open_or_fd("ark:| sleep 0.4; cat > %s" % out, "wb")
    K.write_vec_flt(w, np.arange(3, dtype=np.float32), key="u")
    w.close()
    assert out.exists() and out.stat().st_size == 2 + 6 + 4 + 12        # complete the moment close() returns, no polling
    w.close()                                                            # idempotent
    bad = K.open_or_fd("ark:| cat > /dev/null; exit 7", "wb")
    K.write_vec_flt(bad, np.arange(3, dtype=np.float32), key="u")
    with pytest.raises(K.SubprocessFailed):
        bad.close()


REF_KALDI_IO = "/root/reference/pytorch/libs/support/kaldi_io.py"


@pytest.mark.skipif(not os.path.exists(REF_KALDI_IO), reason="reference tree only exists in the build container")
def test_wire_format_against_the_reference_kaldi_io(tmp_path):
    """pin libs.support.kaldi_io to the reference's own vendored module (kaldi_io.py:329-608), in a
    subprocess that imports it from /root/reference: bytes written by either side are equal, and each side reads what
    the other wrote - float / double matrices and vectors, text matrices, and the 'CM ' decode."""
    import subprocess
    import sys
    rng = np.random.RandomState(7)
    cols = 5
    pct = np.sort(rng.randint(0, 65536, size=(cols, 4)), axis=1)
    cm = _cm_entry("cm", rng.randint(0, 256, size=(cols, 33)), pct, -1.5, 4.25)
    (tmp_path / "cm.ark").write_bytes(cm)
    (tmp_path / "txt.ark").write_bytes(b"tx  [\n  1.5 -2 3\n  4 5 6.25 ]\n")
    np.save(tmp_path / "m32.npy", rng.randn(9, 4).astype(np.float32))
    np.save(tmp_path / "m64.npy", rng.randn(3, 7))
    np.save(tmp_path / "v32.npy", rng.randn(192).astype(np.float32))
    code = r'''
import sys, os, importlib.util, numpy as np
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_kaldi_io", %(ref)r)
R = importlib.util.module_from_spec(spec); spec.loader.exec_module(R)
d = %(d)r
m32, m64, v32 = np.load(d + "/m32.npy"), np.load(d + "/m64.npy"), np.load(d + "/v32.npy")
with open(d + "/ref_out.ark", "wb") as f:
    R.write_mat(f, m32, key="a"); R.write_mat(f, m64, key="b"); R.write_vec_flt(f, v32, key="v")
np.save(d + "/ref_cm.npy", dict(R.read_mat_ark(d + "/cm.ark"))["cm"])
np.save(d + "/ref_txt.npy", dict(R.read_mat_ark(d + "/txt.ark"))["tx"])
# the reference reads what this repo wrote
got = {}
with open(d + "/mine_out.ark", "rb") as f:
    k = R.read_key(f); got[k] = R.read_mat(f)
    k = R.read_key(f); got[k] = R.read_mat(f)
    k = R.read_key(f); got[k] = R.read_vec_flt(f)
assert np.array_equal(got["a"], m32) and np.array_equal(got["b"], m64) and np.array_equal(got["v"], v32)
print("REF-OK")
''' % dict(ref=REF_KALDI_IO, d=str(tmp_path))
    m32, m64, v32 = np.load(tmp_path / "m32.npy"), np.load(tmp_path / "m64.npy"), np.load(tmp_path / "v32.npy")
    with open(tmp_path / "mine_out.ark", "wb") as f:
        kaldi_io.write_mat(f, m32, key="a"); kaldi_io.write_mat(f, m64, key="b"); kaldi_io.write_vec_flt(f, v32, key="v")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPYCACHEPREFIX="/tmp/pyc_ref")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "REF-OK" in out.stdout, out.stdout + out.stderr
    assert (tmp_path / "ref_out.ark").read_bytes() == (tmp_path / "mine_out.ark").read_bytes()         # bytes equal
    assert np.array_equal(dict(kaldi_io.read_mat_ark(str(tmp_path / "cm.ark")))["cm"], np.load(tmp_path / "ref_cm.npy"))   # 'CM ' decode equal
    assert np.array_equal(dict(kaldi_io.read_mat_ark(str(tmp_path / "txt.ark")))["tx"], np.load(tmp_path / "ref_txt.npy"))
    rd = kaldi_io.PackedArkReader(open(tmp_path / "cm.ark", "rb"))
    assert rd.peek_dim() == cols
    buf = np.empty((64, cols), np.float32)
    keys, offs, n = rd.read_group(buf)
    assert keys == ["cm"] and np.array_equal(buf[:n], np.load(tmp_path / "ref_cm.npy"))


def test_matrix_rows_reads_compressed_headers_and_scp_ranges(tmp_path):
    """the length pass of --sharded reads rows from the header of FM / CM / CM2 / CM3 matrices (the compressed
    formats share the GlobalHeader) and honours Kaldi's scp range specifiers."""
    import struct
    import importlib.util
    spec = importlib.util.spec_from_file_location("extract_embeddings_script", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "asv-subtools_amd", "pytorch",
                                                                                          "pipeline", "onestep", "extract_embeddings.py"))
    X = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(X)
    from libs.support import kaldi_io as K
    m = np.arange(35, dtype=np.float32).reshape(7, 5)
    fm = tmp_path / "fm.ark"
    with open(fm, "wb") as f:
        f.write(b"u ")
        off = f.tell()
        K.write_mat(f, m)
    rx = "%s:%d" % (fm, off)
    assert X.matrix_rows(rx) == 7
    assert X.matrix_rows(rx + "[2:4]") == 3 and X.matrix_rows(rx + "[:3]") == 4 and X.matrix_rows(rx + "[5:]") == 2
    assert np.array_equal(X.read_matrix(rx + "[2:4]"), m[2:5]) and np.array_equal(X.read_matrix(rx + "[1:2,0:1]"), m[1:3, 0:2])
    for tag in (b"CM ", b"CM2 ", b"CM3 "):
        p = tmp_path / ("c%d.ark" % len(tag))
        with open(p, "wb") as f:
            f.write(b"u ")
            off = f.tell()
            f.write(b"\0B" + tag + struct.pack("<ffii", 0.0, 1.0, 11, 4))
        assert X.matrix_rows("%s:%d" % (p, off)) == 11
        assert X.matrix_rows("%s:%d[3:9]" % (p, off)) == 7


def test_read_mat_decodes_the_header_only_compressed_formats(tmp_path):
    """matrix_rows accepted CM2 / CM3 entries that read_mat then refused on one rank during the load.  Both are decoded
    now (Kaldi compressed-matrix.cc: kTwoByte / kOneByte - row-major integers behind the GlobalHeader, value = min + range * u / top);
    the reference's reader stops at an assert for them (kaldi_io.py:531)."""
    import struct
    from libs.support import kaldi_io as K
    r = np.random.RandomState(5)
    for tag, dt, top in ((b"CM2 ", np.uint16, 65535.0), (b"CM3 ", np.uint8, 255.0)):
        q = r.randint(0, int(top) + 1, size=(6, 5)).astype(dt)
        gmin, grange = np.float32(-3.25), np.float32(7.5)
        p = tmp_path / ("m_%s.ark" % tag.decode().strip())
        with open(p, "wb") as f:
            f.write(b"utt ")
            off = f.tell()
            f.write(b"\0B" + tag + struct.pack("<ffii", gmin, grange, 6, 5) + q.tobytes())
        want = (gmin + q.astype(np.float32) * np.float32(np.float64(grange) / top)).astype(np.float32)
        got = K.read_mat("%s:%d" % (p, off))
        assert got.dtype == np.float32 and got.shape == (6, 5)
        assert np.allclose(got, want, rtol=0, atol=1e-6)
        assert got.min() >= gmin - 1e-6 and got.max() <= gmin + grange + 1e-6
        ((key, m),) = list(K.read_mat_ark(str(p)))
        assert key == "utt" and np.array_equal(m, got)


def test_scp_batch_loader_packs_a_batch_like_read_mat(tmp_path, monkeypatch):
    """pipeline/onestep/extract_embeddings.py ScpBatchLoader (round 4): the host side of the sharded path reads the utterances of
    a batch straight into one packed buffer - plain float32 entries by header pread + payload preadv on worker threads, every other
    kind of entry (float64, compressed, range specifiers) through read_matrix.  Byte for byte the matrices read_matrix returns, in
    batch order, across two ark files; consecutive batches use alternating buffers (batch k stays intact while k + 1 is loaded)."""
    import importlib.util
    from libs.support import kaldi_io
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("extract_embeddings_mod", os.path.join(repo, "asv-subtools_amd", "pytorch", "pipeline", "onestep",
                                                                                         "extract_embeddings.py"))
    ee = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ee)
    rs = np.random.RandomState(4)
    entries = []
    for a in range(2):
        path = tmp_path / ("feats%d.ark" % a)
        with open(path, "wb") as f:
            for i in range(40):
                key = "a%d_u%03d" % (a, i)
                m = rs.randn(int(rs.randint(1, 60)), 24)
                f.write((key + " ").encode())
                pos = f.tell()
                if i % 11 == 5:
                    kaldi_io.write_mat(f, m.astype(np.float64))                   # 'DM '
                else:
                    kaldi_io.write_mat(f, m.astype(np.float32))
                entries.append((key, "%s:%d" % (path, pos)))
    big = [i for i, (k, rx) in enumerate(entries) if ee.matrix_rows(rx) >= 12][:3]
    for i in big:
        entries[i] = (entries[i][0], entries[i][1] + "[2:9]")                    # Kaldi range specifier: rows 2..9
    want = [ee.read_matrix(rx) for _, rx in entries]
    order = list(rs.permutation(len(entries)))
    from libs.support import native_io
    assert native_io.lib() is not None and native_io.lib().asv_io_version() >= 2, "libasv_io.so is built by `make -C asv-subtools_amd/csrc` (build())"
    for threads, native in ((1, True), (4, True), (1, False)):             # native reads (libasv_io.so) / the Python fallback: the same bytes
        monkeypatch.setattr(native_io, "_LIB", None if native else False)
        ld = ee.ScpBatchLoader(entries, threads=threads)
        try:
            first = ld.load_batch(order[:37])
            keep = [m.copy() for m in first]
            second = ld.load_batch(order[37:])
            for got, idx in ((first, order[:37]), (second, order[37:])):
                assert got.packed.flags["C_CONTIGUOUS"] and got.packed.dtype == np.float32
                assert list(got.offsets) == list(np.concatenate([[0], np.cumsum([want[i].shape[0] for i in idx])]))
                for m, i in zip(got, idx):
                    assert m.shape == want[i].shape and np.array_equal(m, want[i])
            assert all(np.array_equal(a, b) for a, b in zip(first, keep))          # the first batch survived the second load
            assert np.array_equal(ld(order[0]), want[order[0]])
            assert len(ld._fds) == 2                                                # one descriptor per ark file
        finally:
            ld.close()
    # an entry whose matrix runs past the end of its archive: loud in both paths
    with open(tmp_path / "cut.ark", "wb") as f:
        f.write(b"k ")
        pos = f.tell()
        kaldi_io.write_mat(f, rs.randn(30, 24).astype(np.float32))
        f.truncate(f.tell() - 100)
    for native in (True, False):
        monkeypatch.setattr(native_io, "_LIB", None if native else False)
        bad = ee.ScpBatchLoader([("k", "%s:%d" % (tmp_path / "cut.ark", pos))])
        with pytest.raises(kaldi_io.BadInputFormat):
            bad.load_batch([0])
        bad.close()


def _load_extract_script():
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("extract_embeddings_mod", os.path.join(repo, "asv-subtools_amd", "pytorch", "pipeline", "onestep",
                                                                                         "extract_embeddings.py"))
    ee = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ee)
    return ee


def _write_many_arks(tmp_path, n_files, per_file, dim=8, seed=7):
    from libs.support import kaldi_io
    rs = np.random.RandomState(seed)
    entries, want = [], []
    for a in range(n_files):
        path = tmp_path / ("f%03d.ark" % a)
        with open(path, "wb") as f:
            for i in range(per_file):
                key = "f%03d_u%02d" % (a, i)
                m = rs.randn(int(rs.randint(1, 20)), dim).astype(np.float64 if (a + i) % 9 == 4 else np.float32)
                f.write((key + " ").encode())
                pos = f.tell()
                kaldi_io.write_mat(f, m)
                entries.append((key, "%s:%d" % (path, pos)))
                want.append(m.astype(np.float32))
    return entries, want


def test_scp_batch_loader_bounds_its_open_descriptors(tmp_path):
    """a feats.scp over more ark files than descriptors may stay open (augmented / nj-split feature directories against
    the soft RLIMIT_NOFILE): the loader keeps an LRU of `max_open` descriptors, never evicts a file of the batch it is reading,
    and lengths() + every batch still return the right bytes."""
    ee = _load_extract_script()
    entries, want = _write_many_arks(tmp_path, n_files=40, per_file=2)
    ld = ee.ScpBatchLoader(entries, threads=2, max_open=5)
    try:
        assert list(ld.lengths()) == [m.shape[0] for m in want]
        assert len(ld._fds) <= 5
        order = list(np.random.RandomState(1).permutation(len(entries)))
        for lo in range(0, len(order), 24):                  # 24 utterances from up to 24 files per batch: more than max_open at once
            idx = order[lo:lo + 24]
            got = ld.load_batch(idx)
            for m, i in zip(got, idx):
                assert np.array_equal(m, want[i])
            assert len(ld._fds) <= max(5, len({entries[i][1].rsplit(":", 1)[0] for i in idx}))
        one = ld.load_batch([3])
        assert np.array_equal(one[0], want[3]) and len(ld._fds) <= 5 + 1
    finally:
        ld.close()
    assert len(ld._fds) == 0


def test_scp_batch_loader_header_table_equals_the_per_entry_path(tmp_path):
    """Round 5: index_all() reads every plain entry's 15-byte header in ONE native call and load_batch() then builds a batch's read
    list by array indexing (no per-utterance Python).  Same lengths, same bytes as the per-entry path - on a table with float64
    entries and a range specifier in between (not in the table: a batch that holds one goes the per-entry way), with a batch of
    different widths refused, and not at all (False, per-entry path) when the files outnumber `max_open` or an offset points at the
    end of its file."""
    from libs.support import kaldi_io
    ee = _load_extract_script()
    rs = np.random.RandomState(9)
    entries, want = [], []
    for a in range(3):
        path = tmp_path / ("t%d.ark" % a)
        with open(path, "wb") as f:
            for i in range(30):
                key = "t%d_u%02d" % (a, i)
                m = rs.randn(int(rs.randint(1, 50)), 16).astype(np.float32)
                f.write((key + " ").encode())
                pos = f.tell()
                kaldi_io.write_mat(f, m.astype(np.float64) if i % 13 == 4 else m)
                entries.append((key, "%s:%d" % (path, pos)))
                want.append(m)
    entries[7] = (entries[7][0], entries[7][1] + "[0:0]")
    want[7] = want[7][0:1]
    fast, slow = ee.ScpBatchLoader(entries, threads=3), ee.ScpBatchLoader(entries, threads=3)
    slow._table = False
    try:
        assert fast.index_all() is True and slow.index_all() is False
        plain = fast._table[0]
        assert [i for i in range(len(entries)) if not plain[i]] == sorted({7} | {i for i in range(len(entries)) if i % 30 % 13 == 4})      # the range entry, the float64 ones
        assert list(fast.lengths()) == list(slow.lengths()) == [m.shape[0] for m in want]
        order = list(rs.permutation(len(entries)))
        all_plain = [i for i in order if plain[i]]
        for idx in (order[:25], order[25:60], all_plain[:40], all_plain[40:], [all_plain[0]]):
            a, b = fast.load_batch(idx), slow.load_batch(idx)
            assert len(a) == len(b) == len(idx) and list(a.offsets) == list(b.offsets)
            assert np.array_equal(a.packed, b.packed)
            for k, i in enumerate(idx):
                assert np.array_equal(a[k], want[i]) and np.array_equal(a[k - len(idx)], want[i])
            assert [m.shape for m in a] == [want[i].shape for i in idx] and len(a[1:3]) == len(idx[1:3])
        with pytest.raises(IndexError):
            a[len(idx)]
    finally:
        fast.close()
        slow.close()
    # another width in the table: a batch mixing the two is refused, by either path
    path = tmp_path / "wide.ark"
    with open(path, "wb") as f:
        f.write(b"w ")
        pos = f.tell()
        kaldi_io.write_mat(f, rs.randn(5, 20).astype(np.float32))
    mixed = entries + [("w", "%s:%d" % (path, pos))]
    ld = ee.ScpBatchLoader(mixed, threads=2)
    try:
        assert ld.index_all()
        with pytest.raises(ValueError, match="different widths"):
            ld.load_batch([0, len(mixed) - 1])
    finally:
        ld.close()
    few = ee.ScpBatchLoader(entries, threads=2, max_open=2)             # 3 files > 2 descriptors: per-entry path
    try:
        assert few.index_all() is False and list(few.lengths()) == [m.shape[0] for m in want]
    finally:
        few.close()
    size = os.path.getsize(tmp_path / "t0.ark")
    at_end = ee.ScpBatchLoader(entries + [("bad", "%s:%d" % (tmp_path / "t0.ark", size - 3))], threads=2)
    try:
        assert at_end.index_all() is False                                 # (the per-entry path reports which entry it is)
    finally:
        at_end.close()


def test_scp_batch_loader_fills_the_callers_buffers(tmp_path):
    """The sharded path hands the loader the page-locked input buffers of the device pipeline (libs.amd.pipeline.DeviceSets): batches
    land in them alternately, `before_fill(turn)` is called before a buffer is overwritten, and a batch that does not fit gets an
    array of its own (`own_array`; its slot of the rotation is consumed all the same)."""
    ee = _load_extract_script()
    entries, want = _write_many_arks(tmp_path, n_files=3, per_file=12)
    bufs = [np.full((80, 8), np.nan, dtype=np.float32) for _ in range(2)]
    calls = []
    ld = ee.ScpBatchLoader(entries, threads=2, buffers=bufs, before_fill=calls.append)
    try:
        a = ld.load_batch([0, 1, 2])
        b = ld.load_batch([5, 4])
        assert (a.turn, b.turn) == (0, 1) and calls == [0, 1]
        assert a.packed.base is bufs[0] or a.packed.ctypes.data == bufs[0].ctypes.data
        assert b.packed.ctypes.data == bufs[1].ctypes.data
        for got, idx in ((a, [0, 1, 2]), (b, [5, 4])):
            for m, i in zip(got, idx):
                assert np.array_equal(m, want[i])
        everything = list(range(len(entries)))                # > 80 rows: cannot fit a buffer
        assert sum(m.shape[0] for m in want) > 80
        c = ld.load_batch(everything)
        assert c.own_array and c.turn == 0 and calls == [0, 1, 0]        # the slot of the rotation is consumed all the same
        assert all(np.array_equal(m, want[i]) for m, i in zip(c, everything))
        d = ld.load_batch([7])
        assert d.turn == 1 and not d.own_array and not a.own_array
    finally:
        ld.close()


def test_scp_group_reader_streams_the_table_in_order(tmp_path):
    """`scp:` input without --sharded goes through extract_stream over ScpGroupReader (PackedArkReader's interface): groups in scp
    order bounded by the buffer and by max_utts, an utterance longer than the buffer alone in its own array, float64 entries
    converted - every matrix exactly once, byte for byte what read_matrix returns."""
    ee = _load_extract_script()
    from libs.support import kaldi_io
    entries, want = _write_many_arks(tmp_path, n_files=4, per_file=9)
    with open(tmp_path / "long.ark", "wb") as f:              # one utterance longer than the whole buffer, in the middle of the table
        f.write(b"long ")
        pos = f.tell()
        big = np.random.RandomState(2).randn(70, 8).astype(np.float32)
        kaldi_io.write_mat(f, big)
    entries.insert(17, ("long", "%s:%d" % (tmp_path / "long.ark", pos)))
    want.insert(17, big)
    rd = ee.ScpGroupReader(entries, threads=2)
    try:
        assert rd.peek_dim() == 8
        buf = np.empty((50, 8), dtype=np.float32)
        seen = []
        while True:
            keys, offs, frames = rd.read_group(buf, max_utts=7)
            if not keys:
                break
            assert len(keys) <= 7 and len(offs) == len(keys) + 1
            data = frames if isinstance(frames, np.ndarray) else buf[:frames]
            for j, k in enumerate(keys):
                seen.append((k, data[offs[j]:offs[j + 1]].copy()))
        assert [k for k, _ in seen] == [k for k, _ in entries]
        assert all(np.array_equal(m, w) for (_, m), w in zip(seen, want))
        assert rd.peek_dim() is None
    finally:
        rd.close()
    # a width change inside the table is an error that names the entry
    with open(tmp_path / "wide.ark", "wb") as f:
        f.write(b"wide ")
        pos = f.tell()
        kaldi_io.write_mat(f, np.zeros((3, 9), dtype=np.float32))
    rd = ee.ScpGroupReader(entries[:2] + [("wide", "%s:%d" % (tmp_path / "wide.ark", pos))])
    with pytest.raises(kaldi_io.BadInputFormat, match="wide"):
        while rd.read_group(np.empty((500, 8), dtype=np.float32), 64)[0]:
            pass
    rd.close()


def test_read_scp_accepts_the_rspecifier_forms_of_the_reference(tmp_path, monkeypatch):
    """'scp:file', 'scp,p:file' / 'scp,s,cs:file' (options accepted and ignored, like read_mat_scp's open_or_fd), no
    prefix, 'scp:cmd |' and 'scp:-'."""
    import io
    ee = _load_extract_script()
    scp = tmp_path / "feats.scp"
    scp.write_text("utt1 /data/a.ark:12\nutt2 /data/a.ark:3456[0:9]\n\nutt3 gunzip -c /data/b.ark.gz |\n")
    want = [("utt1", "/data/a.ark:12"), ("utt2", "/data/a.ark:3456[0:9]"), ("utt3", "gunzip -c /data/b.ark.gz |")]
    for spec in ("scp:%s", "scp,p:%s", "scp,s,cs:%s", "%s", "scp:cat %s |"):
        assert ee.read_scp(spec % scp) == want, spec
    monkeypatch.setattr(sys, "stdin", io.StringIO(scp.read_text()))
    assert ee.read_scp("scp:-") == want
    assert ee._SCP_PREFIX.match("scp,p:x") and not ee._SCP_PREFIX.match("ark:x")


def test_sharded_mode_wants_a_master_port_from_a_real_launcher(tmp_path, monkeypatch, capsys):
    """a free local port is picked only for the launcher-less single-rank run; WORLD_SIZE > 1 without MASTER_PORT is an
    immediate error (each rank picking its own port would hang until the rendezvous times out).  Reaches no device: the check
    sits in front of init_process_group."""
    ee = _load_extract_script()
    import torch
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: None)
    with pytest.raises(SystemExit):
        ee.main(["--model-blueprint", "x.py", "--model-creation", "X()", "--sharded", "true", "--gpu-id", "0", "m.params", "scp:" + str(tmp_path / "f.scp"), "ark:/dev/null"])
    assert "needs MASTER_PORT" in capsys.readouterr().err


def test_libasv_io_exports_what_its_header_declares():
    import ctypes, re
    from libs.support import native_io
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(repo, "include", "asv_io.h")).read()
    names = sorted(set(re.findall(r"\b(asv_io_[a-z_]+)\s*\(", header)))
    assert names == ["asv_io_last_errno", "asv_io_pack_vec_ark", "asv_io_parse_scp", "asv_io_pread_batch", "asv_io_scan_ark", "asv_io_version"]
    L = ctypes.CDLL(native_io.library_path())
    for n in names:
        assert hasattr(L, n), n


def _drain(reader, cap_rows, dim, max_utts):
    """Everything a packed reader yields: [(keys, offsets, frames, data copy)]."""
    out = []
    buf = np.empty((cap_rows, dim), dtype=np.float32)
    while True:
        keys, offs, frames = reader.read_group(buf, max_utts)
        if not keys:
            return out
        data = frames.copy() if isinstance(frames, np.ndarray) else buf[:frames].copy()
        out.append((list(keys), [int(o) for o in offs], data))


def test_indexed_ark_reader_yields_what_the_sequential_reader_yields(tmp_path):
    """kaldi_io.IndexedArkReader (round 4: native header scan + one batched positioned read per group, libasv_io.so) against
    PackedArkReader on the same files: same groups, same offsets, same bytes - ragged lengths, groups cut by capacity and by
    max_utts, an utterance longer than the buffer (returned alone), entries of other kinds in the middle of the file (float64,
    compressed: the generic reader takes over at that byte), a file opened at an offset, an empty file; a wrong-width entry and a
    truncated archive are loud."""
    rs = np.random.RandomState(11)
    dim = 20

    def write(path, kinds, rows_of):
        with open(path, "wb") as f:
            for i, kind in enumerate(kinds):
                m = rs.randn(rows_of(i), dim)
                f.write(("utt_%04d " % i).encode())
                if kind == "f8":
                    kaldi_io.write_mat(f, m.astype(np.float64))
                elif kind == "cm":
                    kaldi_io.write_mat(f, m.astype(np.float32), compressed=True) if "compressed" in kaldi_io.write_mat.__code__.co_varnames else kaldi_io.write_mat(f, m.astype(np.float64))
                else:
                    kaldi_io.write_mat(f, m.astype(np.float32))

    plain = tmp_path / "plain.ark"
    write(plain, ["f4"] * 300, lambda i: 1 + (i * 37) % 90)
    mixed = tmp_path / "mixed.ark"
    write(mixed, ["f4"] * 40 + ["f8"] + ["f4"] * 25 + ["cm"] + ["f4"] * 10, lambda i: 5 + (i * 13) % 60)
    big = tmp_path / "big.ark"
    write(big, ["f4"] * 12, lambda i: 700 if i in (0, 5) else 30)
    for path in (plain, mixed, big):
        for cap, mx in ((400, 1024), (256, 7), (90, 3)):
            with open(path, "rb") as a, open(path, "rb") as b:
                fast = kaldi_io.IndexedArkReader.open(a)
                assert fast is not None
                slow = kaldi_io.PackedArkReader(b)
                assert fast.peek_dim() == slow.peek_dim() == dim
                got, want = _drain(fast, cap, dim, mx), _drain(slow, cap, dim, mx)
            assert sum(len(g[0]) for g in got) == {plain: 300, mixed: 77, big: 12}[path]
            if path != mixed:
                assert len(got) == len(want)
                for g, w in zip(got, want):
                    assert g[0] == w[0] and g[1] == w[1] and np.array_equal(g[2], w[2])
            else:          # a group ends where the generic reader takes over: same utterances in the same order, cut into groups differently
                flat = lambda gs: [(k, g[2][g[1][j]:g[1][j + 1]]) for g in gs for j, k in enumerate(g[0])]
                for (ka, ma), (kb, mb) in zip(flat(got), flat(want)):
                    assert ka == kb and np.array_equal(ma, mb)
    # opened behind the first entries (an 'ark:file:offset' rspecifier)
    with open(plain, "rb") as a, open(plain, "rb") as b:
        first = kaldi_io.PackedArkReader(open(plain, "rb"))
        first.read_group(np.empty((64, dim), dtype=np.float32), 1)
        skip = len(b"utt_0000 ") + 15 + 1 * dim * 4
        a.seek(skip); b.seek(skip)
        got, want = _drain(kaldi_io.IndexedArkReader.open(a), 300, dim, 64), _drain(kaldi_io.PackedArkReader(b), 300, dim, 64)
        assert got[0][0][0] == "utt_0001" and [g[0] for g in got] == [w[0] for w in want]
    empty = tmp_path / "empty.ark"
    empty.write_bytes(b"")
    with open(empty, "rb") as a:
        rd = kaldi_io.IndexedArkReader.open(a)
        assert rd is not None and rd.peek_dim() is None and rd.read_group(np.empty((8, dim), dtype=np.float32), 4)[0] == []
    # a text archive / a pipe: not this reader's business
    text = tmp_path / "text.ark"
    with open(text, "wb") as f:
        f.write(b"k  [\n 1 2\n 3 4 ]\n")
    with open(text, "rb") as a:
        assert kaldi_io.IndexedArkReader.open(a) is None
    assert kaldi_io.IndexedArkReader.open(io.BytesIO(plain.read_bytes())) is None
    # loud failures: a later entry of another width; the archive cut inside its last matrix
    wide = tmp_path / "wide.ark"
    with open(wide, "wb") as f:
        for i, d in enumerate((dim, dim, dim + 4)):
            f.write(("w%d " % i).encode())
            kaldi_io.write_mat(f, rs.randn(9, d).astype(np.float32))
    with open(wide, "rb") as a:
        rd = kaldi_io.IndexedArkReader.open(a)
        buf = np.empty((100, dim), dtype=np.float32)
        assert rd.read_group(buf, 8)[0] == ["w0", "w1"]
        with pytest.raises(kaldi_io.BadInputFormat):
            rd.read_group(buf, 8)
    cut = tmp_path / "cut2.ark"
    cut.write_bytes(plain.read_bytes()[:-50])
    with open(cut, "rb") as a:
        rd = kaldi_io.IndexedArkReader.open(a)
        with pytest.raises(kaldi_io.BadInputFormat):
            _drain(rd, 4000, dim, 1024)


def test_indexed_reader_is_not_offered_for_gzip_archives(tmp_path):
    import gzip
    p = tmp_path / "f.ark.gz"
    with gzip.open(p, "wb") as f:
        f.write(b"k ")
        kaldi_io.write_mat(f, np.ones((3, 4), dtype=np.float32))
    with kaldi_io.open_or_fd(str(p), "rb") as f:
        assert kaldi_io.IndexedArkReader.open(f) is None
        rd = kaldi_io.PackedArkReader(f)
        assert rd.peek_dim() == 4


def test_native_vector_ark_packing_equals_write_vec_flt(tmp_path):
    """asv_io_pack_vec_ark (round 5: the writer side of the extraction loop) produces, for a whole batch in one call, the bytes
    kaldi_io.write_vec_flt writes entry by entry - keys of different lengths, batches below the native threshold through the Python
    path, a key containing a newline refused by the native path (falls back)."""
    from libs.support import native_io
    rs = np.random.RandomState(5)
    for n in (1, 15, 16, 327):
        keys = ["spk%d-utt%0*d" % (i % 7, 1 + i % 5, i) for i in range(n)]
        v = rs.randn(n, 192).astype(np.float32)
        ref = io.BytesIO()
        ref.mode = "wb"
        for k, row in zip(keys, v):
            kaldi_io.write_vec_flt(ref, row, key=k)
        got = kaldi_io.vec_flt_ark_bytes(keys, v)
        assert isinstance(got, bytes) and got == ref.getvalue(), n
        assert native_io.pack_vec_ark(keys, v) == ref.getvalue()
    odd = ["a\nb"] + ["k%d" % i for i in range(20)]
    v = rs.randn(21, 8).astype(np.float32)
    back = list(kaldi_io.read_vec_flt_ark(io.BytesIO(kaldi_io.vec_flt_ark_bytes(odd[1:], v[1:]))))
    assert [k for k, _ in back] == odd[1:]
    assert kaldi_io.vec_flt_ark_bytes(odd, v).startswith(b"a\nb \0BFV ")


def test_vector_ark_buffer_goes_through_files_gzip_and_pipes(tmp_path):
    """The sharded path's writer hands a segment's packed entries to `w.write()` as the native packer's own uint8 array
    (vec_flt_ark_bytes(as_buffer=True): one copy fewer); whatever open_or_fd returned for the wspecifier - a file, a .gz file, a
    '| cmd' pipe (the reference pipes into copy-vector) - must take it, and the entries must read back (a .gz ark through
    read_vec_flt_ark too: GzipFile.mode is an int, which the reference's read_key trips over)."""
    from libs.support import kaldi_io
    keys = ["k%03d" % i for i in range(40)]
    v = np.random.RandomState(2).randn(40, 16).astype(np.float32)
    blob = kaldi_io.vec_flt_ark_bytes(keys, v, as_buffer=True)
    assert bytes(blob) == kaldi_io.vec_flt_ark_bytes(keys, v)
    targets = {str(tmp_path / "a.ark"): str(tmp_path / "a.ark"), str(tmp_path / "b.ark.gz"): str(tmp_path / "b.ark.gz"),
               "| cat > %s" % (tmp_path / "c.ark"): str(tmp_path / "c.ark")}
    for spec, path in targets.items():
        with kaldi_io.open_or_fd(spec, "wb") as w:
            w.write(blob)
        got = list(kaldi_io.read_vec_flt_ark(path))
        assert [k for k, _ in got] == keys and np.array_equal(np.stack([x for _, x in got]), v), spec


def test_native_scp_parse_matches_the_python_parse(tmp_path):
    """read_scp returns a ScpTable over ONE native parse (libasv_io.so asv_io_parse_scp): the same (key, rxfile) sequence as the per-line
    Python parse for every form a Kaldi table holds - plain 'file:offset', range specifiers, pipes, bare files, tabs, CRLF, blank lines,
    trailing white space, no final newline - and `plain_index()` names exactly the 'path:digits' entries."""
    ee = _load_extract_script()
    from libs.support import native_io
    assert native_io.lib() is not None and native_io.lib().asv_io_version() >= 3
    text = ("utt1 /data/a.ark:11\n"
            "utt2\t/data/a.ark:6411  \r\n"
            "\n"
            "   \n"
            "utt3 /data/b.ark:5[0:9]\n"
            "utt4 gunzip -c /data/c.ark.gz |\n"
            "utt5 /data/plain_matrix.txt\n"
            "utt6 /data/b.ark:123456789012\n"
            "utt7   /data/a.ark:0\n"
            "utt8 /data/with:colon.ark:77\n"
            "utt9 /data/a.ark:12x\n"
            "utt10 C:44")
    path = tmp_path / "t.scp"
    path.write_bytes(text.encode("latin1"))
    table = ee.read_scp("scp:" + str(path))
    assert isinstance(table, ee.ScpTable)
    want = [tuple(line.strip().split(None, 1)) for line in text.splitlines() if line.strip()]
    assert len(table) == len(want) == 10
    assert list(table) == want and [table[i] for i in range(len(want))] == want and table[2:5] == want[2:5] and table[-1] == want[-1]
    assert table.keys() == [k for k, _ in want] and table.keys(3, 6) == [k for k, _ in want[3:6]]
    pid, off, paths = table.plain_index()
    plain = {"utt1": ("/data/a.ark", 11), "utt2": ("/data/a.ark", 6411), "utt6": ("/data/b.ark", 123456789012), "utt7": ("/data/a.ark", 0),
             "utt8": ("/data/with:colon.ark", 77), "utt10": ("C", 44)}
    for i, (k, _) in enumerate(want):
        if k in plain:
            assert pid[i] >= 0 and (paths[pid[i]], int(off[i])) == plain[k], k
        else:
            assert pid[i] == -1, k
    assert paths == ["/data/a.ark", "/data/b.ark", "/data/with:colon.ark", "C"]
    with pytest.raises(IndexError):
        table[10]


def test_loader_over_a_native_table_equals_the_list_path(tmp_path):
    """ScpBatchLoader.index_all over a ScpTable (arrays from the native parse, no per-entry Python) gives the same header table, lengths
    and batches as over a plain list of tuples - incl. entries that are not plain (a range specifier) and a path that does not exist."""
    ee = _load_extract_script()
    entries, want = _write_many_arks(tmp_path, n_files=3, per_file=9)
    entries = list(entries)
    entries.insert(4, ("ranged", entries[2][1] + "[1:3]"))
    scp = tmp_path / "all.scp"
    scp.write_text("".join("%s %s\n" % e for e in entries))
    table = ee.read_scp(str(scp))
    assert isinstance(table, ee.ScpTable) and list(table) == entries
    a, b = ee.ScpBatchLoader(table, threads=2), ee.ScpBatchLoader(entries, threads=2)
    try:
        assert a.index_all() and b.index_all()
        assert np.array_equal(a.lengths(), b.lengths())
        for t, u in zip(a._table[:5], b._table[:5]):
            assert np.array_equal(t, u)
        idx = [0, 3, 4, 5, len(entries) - 1]
        pa, pb = a.load_batch(idx), b.load_batch(idx)
        assert np.array_equal(pa.offsets, pb.offsets) and np.array_equal(pa.packed, pb.packed)
    finally:
        a.close()
        b.close()


def test_bench_loaders_harness_runs_both_header_passes(tmp_path):
    """tools/bench_loaders.py (the host side of --sharded under N concurrent ranks, no device): a tiny table through the shared header pass
    (1 / N of the headers per rank + a shared-memory gather standing in for the all-gather) and through --whole-index - both must account
    for every utterance exactly once across the ranks."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from libs.support import native_io
    if native_io.lib() is None:
        pytest.skip("libasv_io.so not built")
    for extra in ([], ["--whole-index"]):
        r = subprocess.run([sys.executable, os.path.join(repo, "tools", "bench_loaders.py"), "--utts", "600", "--ragged-utts", "0", "--ranks", "1,2", "--threads", "2",
                            "--repeats", "1", "--dir", str(tmp_path / "tables")] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        rows = rec["tables"]["fixed200"]["rows"]
        assert [row["ranks"] for row in rows] == [1, 2]
        for row in rows:
            assert row["best_utts_per_s"] > 0 and set(row["slowest_rank_at_best"]) == {"rank", "seconds", "index", "plan", "read", "pack"}
