// A device allocation that is released on every exit path of the function that made it (the replay entry points
// return early through HIP_TRY).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// A copy that is ordered on the given stream and complete on return.  The plain hipMemcpy runs on the null stream, which waits for
// -- and holds up -- the streams of every other handle of the process: independent runs that share a GPU as host threads (scan
// points, cross-validation units) would take turns instead of running side by side.
static inline hipError_t copy_sync(hipStream_t st, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
