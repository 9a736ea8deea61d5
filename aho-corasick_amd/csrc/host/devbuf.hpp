// Owning device buffer that only ever grows (scratch of the search pipelines).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <vector>

namespace acgpu {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = std::max<size_t>(n, 256);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    template <class T> hipError_t upload(const std::vector<T>& v) {
        hipError_t e = ensure(std::max<size_t>(v.size() * sizeof(T), 16));
        if (e != hipSuccess) return e;
        if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace acgpu
