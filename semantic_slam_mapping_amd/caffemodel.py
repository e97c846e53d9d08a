"""Reader of SegNet driving_webdemo weights from a .caffemodel without Caffe or protobuf (python mirror of include/ssm/caffemodel.h).

The reference loads segnet_weights_driving_webdemo.caffemodel through caffe::Net::CopyTrainedLayersFrom
(/root/reference/src/segnet.cpp:17-23).  The file is a serialized caffe.NetParameter; the protobuf wire format is walked directly
(field numbers: include/ssm/caffemodel.h).  `fold_layers` returns the (weight, scale, shift) triples of ssm_segnet_set_layer with
the batch norm (caffe-segnet "BN": scale, shift; or BVLC "BatchNorm" + "Scale") and the convolution bias folded.
`encode_caffemodel` is the inverse used by the tests (there is no real model file in the reference tree)."""
import struct
import numpy as np

LAYER_NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2",
               "conv5_3", "conv5_3_D", "conv5_2_D", "conv5_1_D", "conv4_3_D", "conv4_2_D", "conv4_1_D", "conv3_3_D", "conv3_2_D", "conv3_1_D",
               "conv2_2_D", "conv2_1_D", "conv1_2_D", "conv1_1_D"]


def _varint(buf, i):
    v = 0; sh = 0
    while True:
        b = buf[i]; i += 1
        v |= (b & 0x7F) << sh
        if not b & 0x80:
            return v, i
        sh += 7


def _fields(buf):
    i = 0; n = len(buf)
    while i < n:
        key, i = _varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v = bytes(buf[i:i + 8]); i += 8
        elif wt == 5:
            v = bytes(buf[i:i + 4]); i += 4
        elif wt == 2:
            ln, i = _varint(buf, i); v = buf[i:i + ln]; i += ln
            if len(v) != ln:
                raise ValueError("caffemodel: truncated length-delimited field")
        else:
            raise ValueError(f"caffemodel: unsupported wire type {wt}")
        yield num, wt, v


def _blob(buf):
    legacy = {}; shape = []; data = []; ddata = []
    for num, wt, v in _fields(buf):
        if 1 <= num <= 4 and wt == 0:
            legacy[num] = v
        elif num == 5 and wt == 2:
            data.append(np.frombuffer(v, "<f4"))
        elif num == 5 and wt == 5:
            data.append(np.frombuffer(v, "<f4"))
        elif num == 8 and wt == 2:
            ddata.append(np.frombuffer(v, "<f8"))
        elif num == 8 and wt == 1:
            ddata.append(np.frombuffer(v, "<f8"))
        elif num == 7 and wt == 2:
            for n2, w2, v2 in _fields(v):
                if n2 == 1 and w2 == 0:
                    shape.append(v2)
                elif n2 == 1 and w2 == 2:
                    j = 0
                    while j < len(v2):
                        d, j = _varint(v2, j); shape.append(d)
    arr = np.concatenate(data) if data else (np.concatenate(ddata).astype(np.float32) if ddata else np.zeros(0, np.float32))
    if not shape and legacy:
        shape = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
    if shape:
        if int(np.prod(shape)) != arr.size:
            raise ValueError("caffemodel: blob shape does not match its data length")
        arr = arr.reshape(shape)
    return arr


def read_caffemodel(path):
    """{layer name: (type, [blob arrays])} for every layer that carries blobs"""
    buf = memoryview(open(path, "rb").read())
    out = {}
    for num, wt, v in _fields(buf):
        if num in (100, 2) and wt == 2:
            v1 = num == 2
            name, typ, blobs = "", "", []
            for n2, w2, v2 in _fields(v):
                if n2 == (4 if v1 else 1) and w2 == 2:
                    name = bytes(v2).decode()
                elif n2 == (5 if v1 else 2):
                    typ = bytes(v2).decode() if w2 == 2 else f"V1:{v2}"
                elif n2 == (6 if v1 else 7) and w2 == 2:
                    blobs.append(_blob(v2))
            if blobs:
                out[name] = (typ, blobs)
    if not out:
        raise ValueError(f"caffemodel: no layer with blobs in {path}")
    return out


def fold_layers(net, shapes=None, bn_eps=1e-5):
    """[(weight[Cout][Cin][3][3], scale[Cout], shift[Cout])] in forward order (formulas: include/ssm/caffemodel.h)"""
    out = []
    for l, name in enumerate(LAYER_NAMES):
        if name not in net:
            raise KeyError(f"caffemodel: layer {name} not found")
        blobs = net[name][1]
        w = np.asarray(blobs[0], np.float32)
        if shapes is not None:
            cin, cout = shapes[l]
            w = w.reshape(cout, cin, 3, 3)
        cout = w.shape[0]
        bias = np.asarray(blobs[1], np.float32).reshape(-1) if len(blobs) > 1 else np.zeros(cout, np.float32)
        scale = np.ones(cout, np.float32); shift = bias.copy()
        bn = net.get(name + "_bn")
        if bn is not None:
            bb = [np.asarray(b, np.float32).reshape(-1) for b in bn[1]]
            if len(bb) == 2:
                scale = bb[0].copy(); shift = (bb[0] * bias + bb[1]).astype(np.float32)
            elif len(bb) == 3:
                fac = np.float32(0.0) if bb[2][0] == 0 else np.float32(1.0) / bb[2][0]
                g = np.ones(cout, np.float32); be = np.zeros(cout, np.float32)
                sc = net.get(name + "_scale")
                if sc is not None:
                    g = np.asarray(sc[1][0], np.float32).reshape(-1)
                    if len(sc[1]) > 1:
                        be = np.asarray(sc[1][1], np.float32).reshape(-1)
                scale = (g / np.sqrt(bb[1] * fac + np.float32(bn_eps))).astype(np.float32)
                shift = (scale * (bias - bb[0] * fac) + be).astype(np.float32)
            else:
                raise ValueError(f"caffemodel: {name}_bn has {len(bb)} blobs")
        out.append((w, scale.astype(np.float32), shift.astype(np.float32)))
    return out


# ---- encoder (tests): a NetParameter holding the given layers
def _enc_varint(v):
    o = bytearray()
    while True:
        b = v & 0x7F; v >>= 7
        if v:
            o.append(b | 0x80)
        else:
            o.append(b); return bytes(o)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_blob(arr, legacy=False, packed=True):
    arr = np.ascontiguousarray(arr, "<f4")
    if legacy:
        dims = (list(arr.shape) + [1, 1, 1, 1])[:4] if arr.ndim <= 4 else list(arr.shape)
        dims = [1] * (4 - arr.ndim) + list(arr.shape) if arr.ndim < 4 else dims
        head = b"".join(_enc_varint((k << 3) | 0) + _enc_varint(int(d)) for k, d in zip((1, 2, 3, 4), dims))
    else:
        head = _ld(7, _ld(1, b"".join(_enc_varint(int(d)) for d in arr.shape)))
    if packed:
        body = _ld(5, arr.tobytes())
    else:
        body = b"".join(_enc_varint((5 << 3) | 5) + struct.pack("<f", float(x)) for x in arr.reshape(-1))
    return head + body


def encode_caffemodel(layers, v1=False, legacy_dims=False):
    """layers: [(name, type, [arrays])] -> bytes of a caffe.NetParameter"""
    out = _ld(1, b"segnet_test")
    for name, typ, blobs in layers:
        if v1:
            body = _ld(4, name.encode()) + _enc_varint((5 << 3) | 0) + _enc_varint(4) + b"".join(_ld(6, _enc_blob(b, legacy=True)) for b in blobs)
            out += _ld(2, body)
        else:
            body = _ld(1, name.encode()) + _ld(2, typ.encode()) + _ld(3, b"x") + b"".join(_ld(7, _enc_blob(b, legacy=legacy_dims)) for b in blobs)
            out += _ld(100, body)
    return out
