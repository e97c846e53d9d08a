// ssm/device.h -- one ssm_ctx per (configuration, host thread).  The reference calls detectFeatures on the main
// thread and generatePointCloud on the viewer thread: each thread gets its own context/stream.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <vector>
#include <string>
#include "../ssm_hip.h"
namespace ssm {
struct DeviceError : std::runtime_error { int code; DeviceError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };
// the GPU every context of this process is created on: one process per GPU in the multi-rank driver (exp_mapping --ranks N sets it to the rank)
inline int& default_device() { static int d = 0; return d; }
// contexts created by this process so far: exp_mapping --ranks forks + execs its ranks and must not have touched HIP before (a forked copy of a process whose
// HIP runtime is initialised is not usable, and an exec from such a process takes the node down on this pool): the driver checks this counter at the fork
inline std::atomic<int>& devices_created() { static std::atomic<int> n{0}; return n; }
// page-locked frame buffers (ssm_host_alloc), recycled: hipHostMalloc costs far more than a frame's processing, so a buffer goes back to a free list when the last cv::Mat
// that wraps it dies.  One pool per process; buffers are grouped by size.
class PinnedPool {
public:
    static PinnedPool& instance() { static PinnedPool* p = new PinnedPool; return *p; }      // (leaked on purpose: frames may die after static destruction has begun)
    std::shared_ptr<void> take(size_t bytes) {
        void* p = nullptr;
        { std::lock_guard<std::mutex> lk(mu); auto& fl = free_[bytes]; if (!fl.empty()) { p = fl.back(); fl.pop_back(); } }
        if (!p && ssm_host_alloc(bytes, &p) != SSM_OK) throw DeviceError(SSM_E_NOMEM, "ssm_host_alloc failed");
        return std::shared_ptr<void>(p, [this, bytes](void* q) { std::lock_guard<std::mutex> lk(mu); free_[bytes].push_back(q); });
    }
private:
    std::mutex mu; std::map<size_t, std::vector<void*>> free_;
};
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
