// ssm/parameter_reader.h -- rgbd_tutor::ParameterReader (reference include/parameter_reader.h:9-67): the same flat
// `key=value` file with '#' comments, the same key names (parameters.txt).  Differences: no Boost (std::istringstream),
// a missing key throws std::out_of_range instead of dereferencing end() (parameter_reader.h:52-61), optional default.
#pragma once
#include "common_headers.h"
#include "utils.h"
namespace rgbd_tutor {
class ParameterReader {
public:
    ParameterReader(string filename = "./parameters.txt") {
        ifstream fin(filename.c_str());
        if (!fin) { fin.open("../parameters.txt"); if (!fin) { cerr << "parameter file does not exist." << endl; return; } }
        string str;
        while (getline(fin, str)) {
            if (!str.empty() && str[0] == '#') continue;
            size_t pos = str.find('#');
            if (pos != string::npos) str = str.substr(0, pos);
            pos = str.find('=');
            if (pos == string::npos) continue;
            string key = str.substr(0, pos), value = str.substr(pos + 1);
            while (!value.empty() && (value.back() == '\r' || value.back() == ' ' || value.back() == '\t')) value.pop_back();
            data[key] = value;
        }
    }
    template <class T> T getData(const string& key) const {
        auto it = data.find(key);
        if (it == data.end()) { cerr << "Parameter name " << key << " not found!" << endl; throw out_of_range("parameter " + key); }
        istringstream ss(it->second); T v{}; ss >> v;
        if (ss.fail()) throw invalid_argument("parameter " + key + " = '" + it->second + "'");
        return v;
    }
    template <class T> T getData(const string& key, const T& dflt) const { return data.count(key) ? getData<T>(key) : dflt; }
    bool has(const string& key) const { return data.count(key) != 0; }
    void set(const string& key, const string& value) { data[key] = value; }       // override a key after loading (tests, drivers); not in the reference
    CAMERA_INTRINSIC_PARAMETERS getCamera() const {           // reference src/parameter_reader.cpp:4-19
        CAMERA_INTRINSIC_PARAMETERS c;
        c.fx = getData<double>("camera.fx"); c.fy = getData<double>("camera.fy"); c.cx = getData<double>("camera.cx"); c.cy = getData<double>("camera.cy");
        c.d0 = getData<double>("camera.d0", 0.0); c.d1 = getData<double>("camera.d1", 0.0); c.d2 = getData<double>("camera.d2", 0.0);
        c.d3 = getData<double>("camera.d3", 0.0); c.d4 = getData<double>("camera.d4", 0.0); c.scale = getData<double>("camera.scale");
        return c;
    }
    // the ssm_config this parameter file implies (frame geometry comes from the data source)
    ssm_config deviceConfig(int width, int height) const {
        ssm_config c; ssm_config_default(&c);
        c.width = width; c.height = height;
        c.orb_features = getData<int>("orb_features", c.orb_features); c.orb_scale = getData<float>("orb_scale", c.orb_scale);
        c.orb_levels = getData<int>("orb_levels", c.orb_levels); c.orb_iniThFAST = getData<int>("orb_iniThFAST", c.orb_iniThFAST);
        c.orb_minThFAST = getData<int>("orb_minThFAST", c.orb_minThFAST); c.knn_match_ratio = getData<double>("knn_match_ratio", c.knn_match_ratio);
        c.tracker_ref_frames = getData<int>("tracker_ref_frames", c.tracker_ref_frames);
        c.mapper_resolution = getData<double>("mapper_resolution", c.mapper_resolution); c.mapper_max_distance = getData<double>("mapper_max_distance", c.mapper_max_distance);
        if (has("camera.fx")) { CAMERA_INTRINSIC_PARAMETERS k = getCamera(); c.camera.cx = k.cx; c.camera.cy = k.cy; c.camera.fx = k.fx; c.camera.fy = k.fy; c.camera.scale = k.scale; }
        c.max_batch = getData<int>("ssm_max_batch", 1); c.voxel_capacity_log2 = getData<int>("ssm_voxel_capacity_log2", 20);
        // implementation knobs a parameter file may set (0 = the library's default): where the map stops growing, the SGBM formulation / streams, stereo pairs per launch
        c.voxel_max_capacity_log2 = getData<int>("ssm_voxel_max_capacity_log2", c.voxel_max_capacity_log2);
        c.sgbm_form = getData<int>("ssm_sgbm_form", c.sgbm_form); c.sgbm_streams = getData<int>("ssm_sgbm_streams", c.sgbm_streams);
        c.stereo_batch = getData<int>("ssm_stereo_batch", c.stereo_batch);
        return c;
    }
    map<string, string> data;
};
}  // namespace rgbd_tutor
