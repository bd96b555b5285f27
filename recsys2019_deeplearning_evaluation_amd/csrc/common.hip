// common.hip -- error state, device binding and the device-query entry points of libmi355rec.so.
#include "common.h"

#include <map>
#include <mutex>
#include <unordered_map>

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

// ---- cache of device blocks (common.h) -----------------------------------------------------------------------------------------
namespace {
struct BlockCache {
    std::mutex lock;
    std::multimap<size_t, void *> free_blocks;       // size -> block (of the device the cache was first used on)
    std::unordered_map<void *, size_t> size_of;      // every block handed out by device_block: its true size
    int device = -1;                                 // one process drives one GPU; blocks of any other device bypass the cache
    size_t cached = 0, limit = 0;
    BlockCache() {
        const char *v = getenv("MI355REC_POOL_BYTES");
        limit = v && *v ? (size_t)strtoull(v, nullptr, 10) : (size_t)8 << 30;
    }
};
BlockCache &block_cache() {
    static BlockCache *c = new BlockCache();         // never destroyed: the HIP runtime may be gone before static destructors run
    return *c;
}
constexpr size_t MAX_CACHED_BLOCK = (size_t)1 << 30;
}  // namespace

void *device_block(size_t bytes) {
    BlockCache &c = block_cache();
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> g(c.lock);
        if (c.device < 0) c.device = dev;
        auto it = c.device == dev ? c.free_blocks.lower_bound(bytes) : c.free_blocks.end();
        if (it != c.free_blocks.end() && it->first <= 2 * bytes + 4096) {
            void *p = it->second;
            c.cached -= it->first;
            c.size_of[p] = it->first;
            c.free_blocks.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                           // out of memory: give the cache back and try once more
        (void)hipGetLastError();
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> g(c.lock);
            for (auto &kv : c.free_blocks) drop.push_back(kv.second);
            c.free_blocks.clear();
            c.cached = 0;
        }
        for (void *q : drop) (void)hipFree(q);
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) fail(MI355REC_E_HIP, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    std::lock_guard<std::mutex> g(c.lock);
    if (c.device == dev) c.size_of[p] = bytes;       // (unknown blocks are simply freed when they come back)
    return p;
}

namespace {
thread_local ReleaseScope *t_release_scope = nullptr;
}
ReleaseScope::ReleaseScope(hipStream_t a, hipStream_t b, hipStream_t c) : streams{a, b, c}, outer(t_release_scope) { t_release_scope = this; }
ReleaseScope::~ReleaseScope() { t_release_scope = outer; }
void ReleaseScope::forget(hipStream_t s) {
    for (ReleaseScope *sc = t_release_scope; sc; sc = sc->outer)
        for (hipStream_t &st : sc->streams)
            if (st == s) st = nullptr;
}

void device_block_return(void *p, size_t) {
    if (!p) return;
    BlockCache &c = block_cache();
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(c.lock);
        auto it = c.size_of.find(p);
        if (it != c.size_of.end()) {
            bytes = it->second;
            c.size_of.erase(it);
        }
    }
    if (bytes == 0 || bytes > MAX_CACHED_BLOCK || c.limit == 0) {
        (void)hipFree(p);
        return;
    }
    if (const ReleaseScope *scope = t_release_scope) {   // the owner's streams: nothing else can hold work on the block
        for (hipStream_t st : scope->streams)
            if (st) (void)hipStreamSynchronize(st);
    } else {
        (void)hipDeviceSynchronize();                // what hipFree would have waited for
    }
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> g(c.lock);
        c.free_blocks.emplace(bytes, p);
        c.cached += bytes;
        // Over the limit: give back down to HALF of it, largest first.  (Stopping at the limit left a cache that a search had once filled
        // -- 32 MF models: 8 GiB of factor and schedule buffers -- sitting exactly there: every block returned afterwards pushed another
        // one out, and a hipFree waits for the device.  The similarity constructor's twenty temporaries cost 5 ms more in such a
        // process: ItemKNN fit 14 ms at the end of bench.py against 8.8 ms in a fresh one.)
        if (c.cached > c.limit)
            while (c.cached > c.limit / 2 && !c.free_blocks.empty()) {
                auto last = std::prev(c.free_blocks.end());
                drop.push_back(last->second);
                c.cached -= last->first;
                c.free_blocks.erase(last);
            }
    }
    for (void *q : drop) (void)hipFree(q);
}

namespace {
struct HandlePool {
    std::mutex lock;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
};
HandlePool &handle_pool() {
    static HandlePool *p = new HandlePool();
    return *p;
}
}  // namespace

hipStream_t pooled_stream() {
    HandlePool &p = handle_pool();
    {
        std::lock_guard<std::mutex> g(p.lock);
        if (!p.streams.empty()) {
            hipStream_t s = p.streams.back();
            p.streams.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    MI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

void pooled_stream_return(hipStream_t s) {
    if (!s) return;
    if (hipStreamSynchronize(s) != hipSuccess) {      // a stream that cannot be drained is not handed to anybody else
        (void)hipGetLastError();
        (void)hipStreamDestroy(s);
        return;
    }
    HandlePool &p = handle_pool();
    std::lock_guard<std::mutex> g(p.lock);
    if (p.streams.size() < 64) p.streams.push_back(s);
    else (void)hipStreamDestroy(s);
}

hipEvent_t pooled_event() {
    HandlePool &p = handle_pool();
    {
        std::lock_guard<std::mutex> g(p.lock);
        if (!p.events.empty()) {
            hipEvent_t e = p.events.back();
            p.events.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    MI_HIP(hipEventCreate(&e));
    return e;
}

void pooled_event_return(hipEvent_t e) {
    if (!e) return;
    HandlePool &p = handle_pool();
    std::lock_guard<std::mutex> g(p.lock);
    if (p.events.size() < 256) p.events.push_back(e);
    else (void)hipEventDestroy(e);
}

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
        // A device-to-device hipMemcpy is queued on the null stream and may return before it has run; the handles' streams are
        // non-blocking, so a kernel launched next on one of them would not wait for it (seen once in ~10 suite runs as a stale exchange
        // slab in tests/test_sharding_gpu.py::test_exact_multi_gpu_bpr_emulated_on_one_gpu).  Every entry point of this library blocks.
        if (to_device == 2) MI_HIP(hipStreamSynchronize(nullptr));
    });
}

extern "C" int mi355rec_device_trim(uint64_t *freed_bytes) {
    return guarded([&] {
        BlockCache &c = block_cache();
        std::vector<void *> drop;
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> g(c.lock);
            for (auto &kv : c.free_blocks) drop.push_back(kv.second);
            bytes = c.cached;
            c.free_blocks.clear();
            c.cached = 0;
        }
        for (void *q : drop) (void)hipFree(q);
        if (freed_bytes) *freed_bytes = (uint64_t)bytes;
    });
}

extern "C" int mi355rec_device_synchronize(void) {
    return guarded([&] {
        ensure_device();
        MI_HIP(hipDeviceSynchronize());
    });
}
