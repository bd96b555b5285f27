// common.hip -- error state, device binding and the device-query entry points of libmi355rec.so.
#include "common.h"

namespace mi355rec {

static thread_local std::string g_last_error;
static int g_device = -1;          // chosen by mi355rec_set_device(); -1 = default (0)
static bool g_device_checked = false;
static int g_cu_count = 0;

void set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }

void fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

void ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        fail(MI355REC_E_NO_DEVICE, "no HIP device visible (%s); libmi355rec.so has no CPU fallback",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    int dev = g_device < 0 ? 0 : g_device;
    if (dev >= n) fail(MI355REC_E_INVALID, "device %d requested but only %d visible", dev, n);
    MI_HIP(hipSetDevice(dev));
    if (!g_device_checked) {
        hipDeviceProp_t prop;
        MI_HIP(hipGetDeviceProperties(&prop, dev));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            fail(MI355REC_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", dev,
                 prop.gcnArchName);
        g_cu_count = prop.multiProcessorCount;
        g_device_checked = true;
    }
}

int multiprocessor_count() { return g_cu_count > 0 ? g_cu_count : 256; }

}  // namespace mi355rec

using namespace mi355rec;

extern "C" const char *mi355rec_last_error(void) { return g_last_error.c_str(); }

extern "C" int mi355rec_device_count(int *count) {
    return guarded([&] {
        MI_REQUIRE(count != nullptr, "count is NULL");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        *count = n;
    });
}

extern "C" int mi355rec_set_device(int device) {
    return guarded([&] {
        MI_REQUIRE(device >= 0, "device must be >= 0");
        g_device = device;
        g_device_checked = false;
    });
}

extern "C" int mi355rec_device_name(char *buf, int buf_len) {
    return guarded([&] {
        MI_REQUIRE(buf != nullptr && buf_len > 0, "bad buffer");
        ensure_device();
        int dev = 0;
        MI_HIP(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        MI_HIP(hipGetDeviceProperties(&prop, dev));
        snprintf(buf, buf_len, "%s", prop.gcnArchName);
    });
}

// ---- raw device buffers for callers that exchange results between GPUs without PyTorch (sharding.py over RCCL through ctypes)
extern "C" int mi355rec_device_malloc(void **out, uint64_t bytes) {
    return guarded([&] {
        MI_REQUIRE(out != nullptr, "out is NULL");
        ensure_device();
        *out = nullptr;
        if (bytes) MI_HIP(hipMalloc(out, (size_t)bytes));
    });
}

extern "C" int mi355rec_device_free(void *p) {
    return guarded([&] {
        if (p) MI_HIP(hipFree(p));
    });
}

extern "C" int mi355rec_device_memcpy(void *dst, const void *src, uint64_t bytes, int to_device) {
    return guarded([&] {
        MI_REQUIRE(dst && src, "NULL argument");
        ensure_device();
        MI_REQUIRE(to_device >= 0 && to_device <= 2, "to_device must be 0, 1 or 2");
        MI_HIP(hipMemcpy(dst, src, (size_t)bytes, to_device == 2 ? hipMemcpyDeviceToDevice : (to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost)));
    });
}

extern "C" int mi355rec_device_synchronize(void) {
    return guarded([&] {
        ensure_device();
        MI_HIP(hipDeviceSynchronize());
    });
}
