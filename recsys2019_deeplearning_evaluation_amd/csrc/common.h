// common.h -- shared host-side helpers of libmi355rec.so (error reporting, device buffers, timing).
// gfx950 only; no CUDA path, no compatibility macros.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi355rec.h"

namespace mi355rec {

// ---- error plumbing --------------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const char *msg);
[[noreturn]] void fail(int code, const char *fmt, ...);

#define MI_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            ::mi355rec::fail(MI355REC_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                             __FILE__, __LINE__);                                                     \
    } while (0)

#define MI_REQUIRE(cond, ...)                                                                         \
    do {                                                                                              \
        if (!(cond)) ::mi355rec::fail(MI355REC_E_INVALID, __VA_ARGS__);                               \
    } while (0)

// Wraps the body of every extern "C" entry point: exceptions never cross the C ABI.
template <class F>
static inline int guarded(F &&f) {
    try {
        f();
        return MI355REC_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return MI355REC_E_HIP;
    }
}

// Lazily binds this process to its device (mi355rec_set_device or device 0) and checks that it is gfx950.
void ensure_device();
int multiprocessor_count();

// ---- device memory ---------------------------------------------------------------------------------
// Blocks go back to a small per-process cache instead of hipFree (which drains the device and costs 0.2 - 0.5 ms per block: a dozen
// temporaries made up a third of the similarity constructor): device_block() hands out a cached block of at least -- and at most
// twice -- the size asked for, or calls hipMalloc.  Like hipFree, returning a block waits first (the block may be handed to another
// stream next): for the streams of the calling thread's ReleaseScope, or -- outside any scope -- for the whole device.  Blocks above 1 GiB are not cached; the cache holds at most MI355REC_POOL_BYTES (default 8 GiB of
// the 288, 0 switches it off); mi355rec_device_trim empties it.
void *device_block(size_t bytes);
void device_block_return(void *p, size_t bytes);

// The device-wide wait of a returned block stands in for "every stream that touched the block is drained".  A handle knows those
// streams: while a ReleaseScope is alive on the calling thread, returned blocks wait for the scope's streams ONLY -- so closing one
// handle (or the temporaries of its constructor) no longer waits for another handle's persistent kernel, nor for handles that run
// in other threads.  Outside any scope the device-wide wait stays.  Scopes nest (the innermost one counts).
struct ReleaseScope {
    hipStream_t streams[3];
    ReleaseScope *outer;
    explicit ReleaseScope(hipStream_t a, hipStream_t b = nullptr, hipStream_t c = nullptr);
    ~ReleaseScope();
    // a handle's destructor has drained `s` and is about to destroy it (or hand it back to the pool): no scope of this thread waits for it again
    static void forget(hipStream_t s);
    ReleaseScope(const ReleaseScope &) = delete;
    ReleaseScope &operator=(const ReleaseScope &) = delete;
};

template <class T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr) device_block_return(ptr, count * sizeof(T));
        ptr = nullptr;
        count = 0;
    }
    void swap(DeviceBuffer &other) {
        std::swap(ptr, other.ptr);
        std::swap(count, other.count);
    }
    void alloc(size_t n) {
        release();
        if (n) ptr = static_cast<T *>(device_block(n * sizeof(T)));
        count = n;
    }
    void alloc_zero(size_t n, hipStream_t s) {
        alloc(n);
        if (n) MI_HIP(hipMemsetAsync(ptr, 0, n * sizeof(T), s));
    }
    void upload(const T *host, size_t n, hipStream_t s) {
        alloc(n);
        if (n) MI_HIP(hipMemcpyAsync(ptr, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *host, size_t n, hipStream_t s) const {
        if (n) MI_HIP(hipMemcpyAsync(host, ptr, n * sizeof(T), hipMemcpyDeviceToHost, s));
    }
};

// ---- streams and events of short-lived handles -----------------------------------------------------------------------
// hipStreamCreate + four hipEventCreate cost 1.8 ms on MI355X / ROCm 7 -- half of a similarity constructor that starts from a
// resident URM, and a search creates one handle per fit.  Handles take their (non-blocking) stream and their timing events from a
// per-process pool and hand them back, drained, when they are destroyed; the pool is never torn down (the HIP runtime may be gone
// before static destructors run).  One pool per process = per device (mi355rec_set_device binds a process to one GPU).
hipStream_t pooled_stream();
void pooled_stream_return(hipStream_t s);
hipEvent_t pooled_event();
void pooled_event_return(hipEvent_t e);

// ---- event-pair timer on a stream -------------------------------------------------------------------
struct StreamTimer {
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool pooled = false;
    void init() {
        MI_HIP(hipEventCreate(&t0));
        MI_HIP(hipEventCreate(&t1));
    }
    void init_pooled() {
        t0 = pooled_event();
        t1 = pooled_event();
        pooled = true;
    }
    void destroy() {
        if (pooled) {
            if (t0) pooled_event_return(t0);
            if (t1) pooled_event_return(t1);
        } else {
            if (t0) (void)hipEventDestroy(t0);
            if (t1) (void)hipEventDestroy(t1);
        }
        t0 = t1 = nullptr;
    }
    void start(hipStream_t s) { MI_HIP(hipEventRecord(t0, s)); }
    void stop(hipStream_t s) { MI_HIP(hipEventRecord(t1, s)); }
    // valid after the stream has been synchronised past stop()
    double elapsed_ms() const {
        float ms = 0.f;
        MI_HIP(hipEventElapsedTime(&ms, t0, t1));
        return ms;
    }
};

// Pool of (start, stop) event pairs attached to individual dispatches (hipExtLaunchKernelGGL).
struct DispatchTimers {
    std::vector<hipEvent_t> start, stop;
    int used = 0;
    void reserve(int n) {
        while ((int)start.size() < n) {
            hipEvent_t a, b;
            MI_HIP(hipEventCreate(&a));
            MI_HIP(hipEventCreate(&b));
            start.push_back(a);
            stop.push_back(b);
        }
    }
    void reset() { used = 0; }
    bool next(hipEvent_t &a, hipEvent_t &b, int limit) {
        if (used >= limit || used >= (int)start.size()) return false;
        a = start[used];
        b = stop[used];
        ++used;
        return true;
    }
    // valid after the stream has been synchronised
    double total_ms() const {
        double t = 0;
        for (int i = 0; i < used; ++i) {
            float ms = 0.f;
            MI_HIP(hipEventElapsedTime(&ms, start[i], stop[i]));
            t += ms;
        }
        return t;
    }
    void destroy() {
        for (auto e : start) (void)hipEventDestroy(e);
        for (auto e : stop) (void)hipEventDestroy(e);
        start.clear();
        stop.clear();
    }
};

static inline int div_up(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace mi355rec
