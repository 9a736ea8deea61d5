// Launch helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <set>
#include <utility>

namespace acgpu {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device (and per loaded module): an automaton may be uploaded
// to several devices from one process (acgpu_upload(aut, device), acgpu_find_overlapping_multi), so the attribute is
// tracked per (device ordinal, kernel) -- under a mutex, searches run concurrently from several host threads.
inline hipError_t ensure_dynamic_lds(const void* func, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, func})) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, func});
    return e;
}

// largest LDS allocation of a workgroup on the current device (cached per ordinal like device_cus below)
inline int device_max_lds() {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (cached[dev] == 0) {
        int v = 0;
        cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) ? v : -1;
    }
    return cached[dev];
}

// compute units of the current device (cached per ordinal: the attribute query costs microseconds per call)
inline int device_cus() {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lk(mu);
    if (cached[dev] == 0) {
        int v = 0;
        cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cached[dev];
}

}  // namespace acgpu
