// ssm/caffemodel.h -- reads the trained weights of SegNet driving_webdemo from a .caffemodel WITHOUT Caffe or libprotobuf.
// The reference loads `segnet_model_driving_webdemo.prototxt` + `segnet_weights_driving_webdemo.caffemodel` through
// caffe::Net::CopyTrainedLayersFrom (/root/reference/src/segnet.cpp:17-23).  A .caffemodel is a serialized caffe.NetParameter
// message; the protobuf wire format (varint / 64-bit / length-delimited / 32-bit fields) is walked directly:
//
//   NetParameter     : 1 name (string) | 100 layer (LayerParameter, repeated) | 2 layers (V1LayerParameter, repeated)
//   LayerParameter   : 1 name | 2 type (string) | 7 blobs (BlobProto, repeated)
//   V1LayerParameter : 4 name | 5 type (enum)   | 6 blobs (BlobProto, repeated)
//   BlobProto        : 1 num 2 channels 3 height 4 width (legacy dims) | 5 data (float, packed or not) | 7 shape (BlobShape) |
//                      8 double_data (double, packed or not)
//   BlobShape        : 1 dim (int64, packed or not)
//
// Layers are found by NAME, those of the driving_webdemo model: conv1_1 .. conv5_3, conv5_3_D .. conv1_1_D, each followed by
// `<conv>_bn` (none after conv1_1_D).  Both batch-norm flavours fold into the (scale, shift) pair of ssm_segnet_set_layer:
//   caffe-segnet "BN" (2 blobs: scale s, shift t; the inference weights carry the statistics folded by compute_bn_statistics.py)
//       y = s * (conv + b) + t                         ->  scale = s,            shift = s * b + t
//   BVLC "BatchNorm" (3 blobs: mean m, variance v, factor f) + optional "Scale" layer `<conv>_scale` (gamma g, beta be):
//       y = g * ((conv + b) - m/f) / sqrt(v/f + eps) + be  ->  scale = g / sqrt(v/f + eps),   shift = scale * (b - m/f) + be
// A missing layer / blob or a shape that is not the driving_webdemo one throws std::runtime_error naming it.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
namespace ssm {
struct CaffeBlob { std::vector<int64_t> shape; std::vector<float> data; size_t count() const { size_t n = 1; for (int64_t d : shape) n *= (size_t)d; return shape.empty() ? data.size() : n; } };
struct CaffeLayer { std::string name, type; std::vector<CaffeBlob> blobs; };

namespace pbwire {
struct Reader {
    const uint8_t* p; const uint8_t* end;
    Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0; int sh = 0;
        while (true) {
            if (p >= end || sh > 63) throw std::runtime_error("caffemodel: truncated varint");
            const uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7;
        }
    }
    // next field: number + wire type; for type 2 `sub` spans the payload
    bool field(uint32_t& num, uint32_t& wt, uint64_t& val, Reader& sub) {
        if (done()) return false;
        const uint64_t key = varint(); num = (uint32_t)(key >> 3); wt = (uint32_t)(key & 7); val = 0;
        switch (wt) {
        case 0: val = varint(); break;
        case 1: if (end - p < 8) throw std::runtime_error("caffemodel: truncated fixed64"); memcpy(&val, p, 8); p += 8; break;
        case 5: { if (end - p < 4) throw std::runtime_error("caffemodel: truncated fixed32"); uint32_t v; memcpy(&v, p, 4); p += 4; val = v; break; }
        case 2: { const uint64_t n = varint(); if ((uint64_t)(end - p) < n) throw std::runtime_error("caffemodel: truncated length-delimited field");
                  sub = Reader(p, (size_t)n); p += n; break; }
        default: throw std::runtime_error("caffemodel: unsupported wire type " + std::to_string(wt));
        }
        return true;
    }
};
inline CaffeBlob parse_blob(Reader r)
{
    CaffeBlob b; int64_t legacy[4] = {-1, -1, -1, -1}; std::vector<double> dd;
    uint32_t num, wt; uint64_t val; Reader sub(nullptr, 0);
    while (r.field(num, wt, val, sub)) {
        if (num >= 1 && num <= 4 && wt == 0) legacy[num - 1] = (int64_t)val;
        else if (num == 5 && wt == 2) { const size_t n = (size_t)(sub.end - sub.p) / 4, o = b.data.size(); b.data.resize(o + n); memcpy(b.data.data() + o, sub.p, n * 4); }
        else if (num == 5 && wt == 5) { uint32_t u = (uint32_t)val; float f; memcpy(&f, &u, 4); b.data.push_back(f); }
        else if (num == 8 && wt == 2) { const size_t n = (size_t)(sub.end - sub.p) / 8, o = dd.size(); dd.resize(o + n); memcpy(dd.data() + o, sub.p, n * 8); }
        else if (num == 8 && wt == 1) { double d; memcpy(&d, &val, 8); dd.push_back(d); }
        else if (num == 7 && wt == 2) {
            uint32_t n2, w2; uint64_t v2; Reader s2(nullptr, 0);
            while (sub.field(n2, w2, v2, s2)) {
                if (n2 == 1 && w2 == 0) b.shape.push_back((int64_t)v2);
                else if (n2 == 1 && w2 == 2) while (!s2.done()) b.shape.push_back((int64_t)s2.varint());
            }
        }
    }
    if (b.data.empty() && !dd.empty()) { b.data.resize(dd.size()); for (size_t i = 0; i < dd.size(); i++) b.data[i] = (float)dd[i]; }
    if (b.shape.empty() && legacy[0] >= 0) for (int i = 0; i < 4; i++) b.shape.push_back(legacy[i] < 0 ? 1 : legacy[i]);
    if (!b.shape.empty() && b.count() != b.data.size()) throw std::runtime_error("caffemodel: blob shape does not match its data length");
    return b;
}
inline CaffeLayer parse_layer(Reader r, bool v1)
{
    CaffeLayer L; uint32_t num, wt; uint64_t val; Reader sub(nullptr, 0);
    const uint32_t f_name = v1 ? 4 : 1, f_type = v1 ? 5 : 2, f_blobs = v1 ? 6 : 7;
    while (r.field(num, wt, val, sub)) {
        if (num == f_name && wt == 2) L.name.assign((const char*)sub.p, (size_t)(sub.end - sub.p));
        else if (num == f_type && wt == 2) L.type.assign((const char*)sub.p, (size_t)(sub.end - sub.p));
        else if (num == f_type && wt == 0) L.type = "V1:" + std::to_string(val);
        else if (num == f_blobs && wt == 2) L.blobs.push_back(parse_blob(sub));
    }
    return L;
}
}  // namespace pbwire

// every layer of the file that carries blobs, by name
inline std::map<std::string, CaffeLayer> read_caffemodel(const std::string& path)
{
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("caffemodel: cannot open " + path);
    const std::streamsize n = in.tellg(); in.seekg(0);
    std::vector<uint8_t> buf((size_t)n);
    if (n > 0 && !in.read((char*)buf.data(), n)) throw std::runtime_error("caffemodel: cannot read " + path);
    std::map<std::string, CaffeLayer> out;
    pbwire::Reader r(buf.data(), buf.size()); uint32_t num, wt; uint64_t val; pbwire::Reader sub(nullptr, 0);
    while (r.field(num, wt, val, sub)) {
        if ((num == 100 || num == 2) && wt == 2) { CaffeLayer L = pbwire::parse_layer(sub, num == 2); if (!L.blobs.empty()) out[L.name] = std::move(L); }
    }
    if (out.empty()) throw std::runtime_error("caffemodel: no layer with blobs in " + path + " (not a caffe NetParameter?)");
    return out;
}

// conv layer names of segnet_model_driving_webdemo.prototxt in forward order = the layer index of ssm_segnet_set_layer
inline const std::vector<std::string>& segnet_layer_names()
{
    static const std::vector<std::string> n = {
        "conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3",
        "conv5_3_D", "conv5_2_D", "conv5_1_D", "conv4_3_D", "conv4_2_D", "conv4_1_D", "conv3_3_D", "conv3_2_D", "conv3_1_D", "conv2_2_D", "conv2_1_D",
        "conv1_2_D", "conv1_1_D"};
    return n;
}
struct FoldedLayer { int cin = 0, cout = 0; std::vector<float> weight, scale, shift; };

// (weight[Cout][Cin][3][3], scale[Cout], shift[Cout]) of conv layer `l`, batch norm and bias folded (formulas in the header comment)
inline FoldedLayer fold_segnet_layer(const std::map<std::string, CaffeLayer>& net, int l, int cin, int cout, float bn_eps = 1e-5f)
{
    const std::string& name = segnet_layer_names().at((size_t)l);
    auto it = net.find(name);
    if (it == net.end() || it->second.blobs.empty()) throw std::runtime_error("caffemodel: layer " + name + " not found");
    const CaffeLayer& conv = it->second;
    FoldedLayer f; f.cin = cin; f.cout = cout;
    if (conv.blobs[0].data.size() != (size_t)cout * cin * 9) throw std::runtime_error("caffemodel: " + name + " weight blob is not " + std::to_string(cout) + "x" + std::to_string(cin) + "x3x3");
    f.weight = conv.blobs[0].data;
    std::vector<float> bias((size_t)cout, 0.f);
    if (conv.blobs.size() > 1) { if (conv.blobs[1].data.size() != (size_t)cout) throw std::runtime_error("caffemodel: " + name + " bias length"); bias = conv.blobs[1].data; }
    f.scale.assign((size_t)cout, 1.f); f.shift = bias;
    auto bn = net.find(name + "_bn");
    if (bn == net.end()) return f;                              // conv1_1_D: plain convolution
    const std::vector<CaffeBlob>& bb = bn->second.blobs;
    auto need = [&](const CaffeBlob& b, const char* what) { if (b.data.size() != (size_t)cout) throw std::runtime_error("caffemodel: " + name + "_bn " + what + " length"); };
    if (bb.size() == 2) {                                       // caffe-segnet BN: scale, shift
        need(bb[0], "scale"); need(bb[1], "shift");
        for (int o = 0; o < cout; o++) { f.scale[o] = bb[0].data[o]; f.shift[o] = bb[0].data[o] * bias[o] + bb[1].data[o]; }
    } else if (bb.size() == 3) {                                // BVLC BatchNorm: mean, variance, moving-average factor (+ Scale layer)
        need(bb[0], "mean"); need(bb[1], "variance");
        if (bb[2].data.empty()) throw std::runtime_error("caffemodel: " + name + "_bn factor blob");
        const float fac = bb[2].data[0] == 0.f ? 0.f : 1.f / bb[2].data[0];
        std::vector<float> g((size_t)cout, 1.f), be((size_t)cout, 0.f);
        auto sc = net.find(name + "_scale");
        if (sc != net.end()) {
            need(sc->second.blobs.at(0), "gamma"); g = sc->second.blobs[0].data;
            if (sc->second.blobs.size() > 1) { need(sc->second.blobs[1], "beta"); be = sc->second.blobs[1].data; }
        }
        for (int o = 0; o < cout; o++) {
            const float s = g[o] / std::sqrt(bb[1].data[o] * fac + bn_eps);
            f.scale[o] = s; f.shift[o] = s * (bias[o] - bb[0].data[o] * fac) + be[o];
        }
    } else throw std::runtime_error("caffemodel: " + name + "_bn has " + std::to_string(bb.size()) + " blobs (expected 2: caffe-segnet BN, or 3: BatchNorm)");
    return f;
}
}  // namespace ssm
