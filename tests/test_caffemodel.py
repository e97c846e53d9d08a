"""Caffe weight ingestion for Classifier() (/root/reference/src/segnet.cpp:17-23 loads prototxt + caffemodel): a synthetic
.caffemodel written by the test encoder (no real model exists in the reference tree) must come back identical through
(a) the python reader and (b) the dependency-free C++ reader (include/ssm/caffemodel.h via host/caffe2ssmw), with the batch norm and the
bias folded into (scale, shift) exactly like the formulas in the header say.  CPU only: no device call."""
import os
import struct
import subprocess
import numpy as np
import pytest

from semantic_slam_mapping_amd import caffemodel as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "semantic_slam_mapping_amd", "host", "caffe2ssmw")


def _net(rng, flavour, small=True):
    """driving_webdemo layer names with small channel counts; returns (encoder layers, expected folded triples)"""
    layers, want = [], []
    chans = [3] + [int(rng.integers(2, 7)) for _ in range(25)] + [12]
    for l, name in enumerate(cm.LAYER_NAMES):
        cin, cout = chans[l], chans[l + 1]
        w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        layers.append((name, "Convolution", [w, b]))
        if name == "conv1_1_D":
            want.append((w, np.ones(cout, np.float32), b)); continue
        if flavour == "segnet_bn":
            s = (1 + 0.1 * rng.standard_normal(cout)).astype(np.float32); t = rng.standard_normal(cout).astype(np.float32)
            layers.append((name + "_bn", "BN", [s.reshape(1, cout, 1, 1), t.reshape(1, cout, 1, 1)]))
            want.append((w, s, (s * b + t).astype(np.float32)))
        else:
            m = rng.standard_normal(cout).astype(np.float32); v = (0.5 + rng.random(cout)).astype(np.float32); f = np.array([0.999], np.float32)
            g = (1 + 0.1 * rng.standard_normal(cout)).astype(np.float32); be = rng.standard_normal(cout).astype(np.float32)
            layers.append((name + "_bn", "BatchNorm", [m, v, f])); layers.append((name + "_scale", "Scale", [g, be]))
            fac = np.float32(1.0) / f[0]
            s = (g / np.sqrt(v * fac + np.float32(1e-5))).astype(np.float32)
            want.append((w, s, (s * (b - m * fac) + be).astype(np.float32)))
        layers.append((name + "_relu", "ReLU", []))          # blob-less layers are skipped by the readers
    return layers, want


def _read_ssmw(path):
    buf = open(path, "rb").read()
    assert buf[:4] == b"SSMW"
    ver, nl = struct.unpack_from("<II", buf, 4); off = 12; out = []
    assert ver == 1
    for _ in range(nl):
        cin, cout = struct.unpack_from("<II", buf, off); off += 8
        w = np.frombuffer(buf, "<f4", cout * cin * 9, off).reshape(cout, cin, 3, 3); off += w.nbytes
        sc = np.frombuffer(buf, "<f4", cout, off); off += 4 * cout
        sh = np.frombuffer(buf, "<f4", cout, off); off += 4 * cout
        out.append((w, sc, sh))
    assert off == len(buf)
    return out


@pytest.mark.parametrize("flavour,kw", [("segnet_bn", {}), ("batchnorm_scale", {}), ("segnet_bn", {"legacy_dims": True}), ("segnet_bn", {"v1": True})])
def test_caffemodel_roundtrip(tmp_path, flavour, kw):
    rng = np.random.default_rng(7)
    layers, want = _net(rng, flavour)
    path = tmp_path / "net.caffemodel"
    path.write_bytes(cm.encode_caffemodel(layers, **kw))
    net = cm.read_caffemodel(str(path))
    assert "conv1_1" in net and "conv1_1_relu" not in net and net["conv3_2"][1][0].shape[-2:] == (3, 3)
    got = cm.fold_layers(net)
    assert len(got) == 26
    for (w, s, t), (ew, es, et) in zip(got, want):
        assert w.tobytes() == ew.tobytes() and s.tobytes() == es.tobytes() and t.tobytes() == et.tobytes()
    # the C++ reader (what Classifier uses) on the same file
    subprocess.run(["make", "-C", os.path.dirname(TOOL), "caffe2ssmw"], check=True, capture_output=True)
    out = tmp_path / "net.ssmw"
    r = subprocess.run([TOOL, str(path), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for (w, s, t), (ew, es, et) in zip(_read_ssmw(str(out)), want):
        assert w.tobytes() == ew.tobytes() and s.tobytes() == es.tobytes() and np.array_equal(t, et)


def test_caffemodel_unpacked_floats_and_errors(tmp_path):
    rng = np.random.default_rng(3)
    layers, want = _net(rng, "segnet_bn")
    # repeated (non-packed) float encoding of one blob, as old protobuf writers emit
    raw = cm.encode_caffemodel(layers[:1])
    name, typ, blobs = layers[0]
    body = cm._ld(1, name.encode()) + cm._ld(2, typ.encode()) + b"".join(cm._ld(7, cm._enc_blob(b, packed=False)) for b in blobs)
    p = tmp_path / "one.caffemodel"; p.write_bytes(cm._ld(100, body))
    net = cm.read_caffemodel(str(p))
    assert net["conv1_1"][1][0].tobytes() == blobs[0].tobytes()
    assert raw != p.read_bytes()
    # missing layers: both readers say which
    with pytest.raises(KeyError, match="conv1_2"):
        cm.fold_layers(net)
    r = subprocess.run([TOOL, str(p), str(tmp_path / "x.ssmw")], capture_output=True, text=True)
    assert r.returncode == 1 and "conv1_2" in r.stderr
    # truncated file
    q = tmp_path / "trunc.caffemodel"; q.write_bytes(cm.encode_caffemodel(layers)[:-7])
    with pytest.raises((ValueError, IndexError)):
        cm.read_caffemodel(str(q))
    r = subprocess.run([TOOL, str(q), str(tmp_path / "y.ssmw")], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr
