"""Host-side plumbing: Kaldi table I/O wire format, nnet.config, blueprint loading."""

import io
import os
import struct

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
