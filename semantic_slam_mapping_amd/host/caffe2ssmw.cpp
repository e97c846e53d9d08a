// caffe2ssmw -- converts a SegNet driving_webdemo .caffemodel into the flat .ssmw file of include/ssm/segnet.h with the
// batch norm and bias folded (include/ssm/caffemodel.h).  Host only: no device call.  Layer shapes are taken from the weight
// blobs ([Cout][Cin][3][3]), so the tool also serves the CPU tests with small synthetic layers.
// usage: caffe2ssmw in.caffemodel out.ssmw
#include "ssm/caffemodel.h"
#include <cstdio>
int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.caffemodel out.ssmw\n", argv[0]); return 2; }
    try {
        const auto net = ssm::read_caffemodel(argv[1]);
        const auto& names = ssm::segnet_layer_names();
        std::ofstream out(argv[2], std::ios::binary);
        if (!out) throw std::runtime_error(std::string("cannot write ") + argv[2]);
        const uint32_t ver = 1, nl = (uint32_t)names.size();
        out.write("SSMW", 4); out.write((const char*)&ver, 4); out.write((const char*)&nl, 4);
        for (int l = 0; l < (int)nl; l++) {
            auto it = net.find(names[l]);
            if (it == net.end()) throw std::runtime_error("layer " + names[l] + " not found");
            const ssm::CaffeBlob& w = it->second.blobs.at(0);
            if (w.shape.size() != 4 || w.shape[2] != 3 || w.shape[3] != 3) throw std::runtime_error(names[l] + ": weight blob is not [Cout][Cin][3][3]");
            const uint32_t cout = (uint32_t)w.shape[0], cin = (uint32_t)w.shape[1];
            const ssm::FoldedLayer f = ssm::fold_segnet_layer(net, l, (int)cin, (int)cout);
            out.write((const char*)&cin, 4); out.write((const char*)&cout, 4);
            out.write((const char*)f.weight.data(), f.weight.size() * 4); out.write((const char*)f.scale.data(), cout * 4); out.write((const char*)f.shift.data(), cout * 4);
        }
        printf("%u layers\n", nl);
    } catch (const std::exception& e) { fprintf(stderr, "caffe2ssmw: %s\n", e.what()); return 1; }
    return 0;
}
