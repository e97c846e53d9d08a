// ssm/segnet.h -- Classifier (reference include/segnet.h:22-46, src/segnet.cpp): SegNet driving_webdemo inference,
// same public interface (`Classifier()`, `Classify(const cv::Mat&, int N = 1) -> vector<Prediction>`), running on the
// MI355X through ssm_segnet_forward instead of Caffe.  The reference compiles the model paths in
// (../models/segnet_model_driving_webdemo.prototxt + segnet_weights_driving_webdemo.caffemodel + semantic12.txt,
// segnet.cpp:17-19): `Classifier()` reads the same three paths, `Classifier(model, trained, labels)` takes them as arguments.
// The .caffemodel is read by ssm/caffemodel.h (protobuf wire format walked directly, BatchNorm + bias folded per layer; no Caffe,
// no libprotobuf); the topology is fixed in the library, so the .prototxt is only checked for existence when it is given.
// A flat binary is accepted in place of the .caffemodel (told by its magic; $SSM_SEGNET_WEIGHTS overrides the default path):
//   "SSMW" u32 version=1 u32 nlayers=26, then per layer: u32 cin, u32 cout, f32 weight[cout][cin][3][3] (Caffe blob
//   order), f32 scale[cout], f32 shift[cout]   (conv bias + BatchNorm folded: y = scale*conv + shift)
// (scripts/export_ssmw.py writes it from numpy arrays).  A missing / malformed weight file throws std::runtime_error where the
// reference aborts through glog CHECK (segnet.cpp:25-30); a missing label file falls back to the 12 driving_webdemo class names.
#pragma once
#include "common_headers.h"
#include "device.h"
#include "caffemodel.h"
typedef std::pair<string, int> Prediction;      // (label, class id), as in the reference (segnet.h:20)
class Classifier {
public:
    Classifier() : Classifier("../models/segnet_model_driving_webdemo.prototxt",
                              getenv("SSM_SEGNET_WEIGHTS") ? getenv("SSM_SEGNET_WEIGHTS") : "../models/segnet_weights_driving_webdemo.caffemodel",
                              "../models/semantic12.txt") {}
    // the reference's three files (src/segnet.cpp:17-19)
    Classifier(const string& model_file, const string& trained_file, const string& label_file, int width = 640, int height = 480) : Classifier(trained_file, label_file, width, height) {
        if (!model_file.empty() && !ifstream(model_file)) throw runtime_error("Classifier: cannot open " + model_file);
    }
    Classifier(const string& weights_file, const string& label_file, int width = 640, int height = 480) : input_geometry_(480, 360) {
        ssm_config cfg; ssm_config_default(&cfg); cfg.width = width; cfg.height = height; cfg.max_batch = 1; cfg.voxel_capacity_log2 = 10;
        dev.reset(new ssm::Device(cfg));
        ifstream in(weights_file, ios::binary);
        if (!in) throw runtime_error("Classifier: cannot open " + weights_file);
        char magic[4] = {0, 0, 0, 0}; uint32_t ver = 0, nl = 0;
        in.read(magic, 4);
        if (in && memcmp(magic, "SSMW", 4) != 0) {              // a .caffemodel (serialized caffe.NetParameter)
            in.close();
            const map<string, ssm::CaffeLayer> net = ssm::read_caffemodel(weights_file);
            for (int l = 0; l < ssm_segnet_num_layers(); l++) {
                int cin, cout; ssm_segnet_layer_shape(l, &cin, &cout, nullptr, nullptr);
                const ssm::FoldedLayer f = ssm::fold_segnet_layer(net, l, cin, cout);
                dev->check(ssm_segnet_set_layer(dev->ctx(), l, f.weight.data(), f.scale.data(), f.shift.data()), "ssm_segnet_set_layer");
            }
        } else {
        in.read((char*)&ver, 4); in.read((char*)&nl, 4);
        if (!in || ver != 1 || (int)nl != ssm_segnet_num_layers()) throw runtime_error("Classifier: bad weight file " + weights_file);
        for (int l = 0; l < (int)nl; l++) {
            uint32_t cin = 0, cout = 0; in.read((char*)&cin, 4); in.read((char*)&cout, 4);
            int ecin, ecout; ssm_segnet_layer_shape(l, &ecin, &ecout, nullptr, nullptr);
            if (!in || (int)cin != ecin || (int)cout != ecout) throw runtime_error("Classifier: layer shape mismatch in " + weights_file);
            vector<float> w((size_t)cout * cin * 9), sc(cout), sh(cout);
            in.read((char*)w.data(), w.size() * 4); in.read((char*)sc.data(), cout * 4); in.read((char*)sh.data(), cout * 4);
            if (!in) throw runtime_error("Classifier: truncated weight file " + weights_file);
            dev->check(ssm_segnet_set_layer(dev->ctx(), l, w.data(), sc.data(), sh.data()), "ssm_segnet_set_layer");
        }
        }
        ifstream labels(label_file);
        string line;
        while (labels && getline(labels, line)) labels_.push_back(line);
        if (labels_.size() < 12) labels_ = {"Sky", "Building", "Pole", "Road Marking", "Road", "Pavement", "Tree", "Sign Symbol", "Fence", "Vehicle", "Pedestrian", "Bike"};
    }
    // one (label, id) pair per net pixel, 360*480 entries row-major (src/segnet.cpp:65-78); N is ignored like in the reference
    std::vector<Prediction> Classify(const cv::Mat& img, int N = 1) {
        (void)N;
        vector<uint8_t> ids((size_t)input_geometry_.width * input_geometry_.height);
        dev->check(ssm_segnet_forward(dev->ctx(), img.data, img.cols, img.rows, (int)img.step, ids.data(), nullptr), "ssm_segnet_forward");
        vector<Prediction> p; p.reserve(ids.size());
        for (uint8_t id : ids) p.push_back(make_pair(labels_[id], (int)id));
        return p;
    }
    // colour-label image at frame size, produced like experiment/segnet.cpp:80-83,131-146
    cv::Mat ColorLabels(const cv::Mat& img) {
        cv::Mat sem(img.rows, img.cols, CV_8UC3);
        dev->check(ssm_segnet_forward(dev->ctx(), img.data, img.cols, img.rows, (int)img.step, nullptr, sem.data), "ssm_segnet_forward");
        return sem;
    }
private:
    unique_ptr<ssm::Device> dev;
    cv::Size input_geometry_;
    std::vector<string> labels_;
};
