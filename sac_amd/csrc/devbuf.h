// A device allocation that is released on every exit path of the function that made it (the replay entry points
// return early through HIP_TRY).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
