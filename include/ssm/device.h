// ssm/device.h -- one ssm_ctx per (configuration, host thread).  The reference calls detectFeatures on the main
// thread and generatePointCloud on the viewer thread: each thread gets its own context/stream.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include "../ssm_hip.h"
namespace ssm {
struct DeviceError : std::runtime_error { int code; DeviceError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };
// the GPU every context of this process is created on: one process per GPU in the multi-rank driver (exp_mapping --ranks N sets it to the rank)
inline int& default_device() { static int d = 0; return d; }
class Device {
public:
    explicit Device(const ssm_config& cfg, int device = -1) {
        if (device < 0) device = default_device();
        int rc = ssm_create(device, &cfg, &ctx_);
        if (rc != SSM_OK) throw DeviceError(rc, std::string("ssm_create: ") + ssm_last_error(nullptr));   // no CPU fallback, fail loudly
    }
    ~Device() { ssm_destroy(ctx_); }
    Device(const Device&) = delete; Device& operator=(const Device&) = delete;
    ssm_ctx* ctx() const { return ctx_; }
    void check(int rc, const char* what) const { if (rc != SSM_OK) throw DeviceError(rc, std::string(what) + ": " + ssm_last_error(ctx_)); }
private:
    ssm_ctx* ctx_ = nullptr;
};
}  // namespace ssm
