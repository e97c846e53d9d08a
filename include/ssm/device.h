// ssm/device.h -- one ssm_ctx per (configuration, host thread).  The reference calls detectFeatures on the main
// thread and generatePointCloud on the viewer thread: each thread gets its own context/stream.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include "../ssm_hip.h"
namespace ssm {
struct DeviceError : std::runtime_error { int code; DeviceError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };
// the GPU every context of this process is created on: one process per GPU in the multi-rank driver (exp_mapping --ranks N sets it to the rank)
inline int& default_device() { static int d = 0; return d; }
// contexts created by this process so far: exp_mapping --ranks forks + execs its ranks and must not have touched HIP before (a forked copy of a process whose
// HIP runtime is initialised is not usable, and an exec from such a process takes the node down on this pool): the driver checks this counter at the fork
inline std::atomic<int>& devices_created() { static std::atomic<int> n{0}; return n; }
class Device {
public:
    explicit Device(const ssm_config& cfg, int device = -1) {
        if (device < 0) device = default_device();
        devices_created()++;
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
