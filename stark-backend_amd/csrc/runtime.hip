// Library/device entry points: init, error names, device memory and streams.
// Thin on purpose: callers that already own a HIP binding pass their own pointers and streams.
// Replaces (reference): crates/cuda-common/src/{common,d_buffer,copy,stream,error}.rs and
// crates/cuda-backend/cuda/src/device_info.cu.
#include "device_ctx.h"
#include "openvm_hip.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <map>
#include <string>
#include <vector>

namespace ovm {
namespace {
struct ProfRec {
    const char *label;
    hipEvent_t a, b;
    int depth;           // number of scopes open on this thread when this one started (0 = a top-level stage)
    hipStream_t stream;  // the stream the events were recorded on
    bool side;           // ... which is the library's side stream of the recording thread
};
thread_local int g_prof_depth = 0;
thread_local hipStream_t g_prof_side = nullptr;
thread_local bool g_prof_has_side = false;
int g_prof_mode = 0;  // 0 off, 1 every stage, 2 the dominant kernel's label only
std::vector<ProfRec> g_prof;
unsigned g_prof_generation = 0;  // bumped whenever the list is cleared (under g_prof_mutex)
std::mutex g_prof_mutex;  // (the engines of a throughput run record from several threads)

// roctx ranges under the same labels (reference: the NVTX layer of crates/stark-sdk/src/nvtx_tracing.rs:9-33).
// OVM_ROCTX=1 in the environment turns them on; the marker library is looked up at run time, so the product has no
// link-time dependency on a profiler (rocprofv3 --marker-trace preloads it and sees the ranges).
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("OVM_ROCTX");
        if (!e || !*e || *e == '0') return;
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr, pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() {
    static Roctx r;
    return r;
}
}  // namespace

namespace {
std::atomic<bool> g_exiting{false};
std::once_flag g_exit_once;
struct ThreadResources {
    std::vector<ThreadResource *> list;
};
ThreadResources &thread_resources() {
    static thread_local ThreadResources r;
    return r;
}
}  // namespace
bool runtime_exiting() { return g_exiting.load(std::memory_order_acquire); }
ThreadResource::ThreadResource() {
    std::call_once(g_exit_once, [] { atexit([] { g_exiting.store(true, std::memory_order_release); }); });
    thread_resources().list.push_back(this);
}
ThreadResource::~ThreadResource() {
    auto &l = thread_resources().list;
    for (size_t i = 0; i < l.size(); i++)
        if (l[i] == this) {
            l.erase(l.begin() + i);
            break;
        }
}

void prof_register_side_stream(hipStream_t s) {
    g_prof_side = s;
    g_prof_has_side = s != nullptr;
}
ProfScope::ProfScope(const char *label, hipStream_t s) : stream(s) {
    if (roctx().push) {
        roctx().push(label);
        ranged = true;
    }
    if (!g_prof_mode) return;
    if (g_prof_mode == 2 && strcmp(label, "merkle.hash_rows") != 0) return;
    ProfRec r{label, nullptr, nullptr, g_prof_depth, s, g_prof_has_side && g_prof_side == s};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    {
        std::lock_guard<std::mutex> lk(g_prof_mutex);
        slot = (int)g_prof.size();
        generation = g_prof_generation;
        g_prof.push_back(r);
    }
    g_prof_depth++;
}
ProfScope::~ProfScope() {
    if (slot >= 0) {
        hipEvent_t b;
        {
            std::lock_guard<std::mutex> lk(g_prof_mutex);
            // profile_enable may have cleared the records while this scope was open -- and another thread may have pushed
            // new ones since: the slot then names somebody else's record, so the end is not recorded at all
            b = generation == g_prof_generation && slot < (int)g_prof.size() ? g_prof[slot].b : nullptr;
        }
        if (b) (void)hipEventRecord(b, stream);
        g_prof_depth--;
    }
    if (ranged) roctx().pop();
}
}  // namespace ovm

extern "C" {

int ovm_hip_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(ovm::g_prof_mutex);
    for (auto &r : ovm::g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    ovm::g_prof.clear();
    ovm::g_prof_generation++;
    ovm::g_prof_mode = on < 0 || on > 2 ? 1 : on;
    return 0;
}

// Writes "label calls total_ms nested_ms side_ms\n" lines (aggregated per label) into buf; returns bytes needed.
// nested_ms: the part of total_ms measured while ANOTHER scope was open on the thread (e.g. the Merkle and NTT labels inside
// whir.rs_commit); side_ms: the part measured on the library's own side stream of the recording thread (the early coset
// LDE beside the GKR layers) -- decided when the scope is recorded, not by which stream happened to be seen first
// (ADVICE r4). total - nested - side summed over the labels is additive: it cannot exceed the step's time.
uint64_t ovm_hip_profile_report(char *buf, uint64_t cap) {
    (void)hipDeviceSynchronize();
    struct Agg {
        uint64_t calls = 0;
        double ms = 0, nested = 0, side = 0;
    };
    std::map<std::string, Agg> agg;
    std::vector<std::string> order;
    std::lock_guard<std::mutex> lk(ovm::g_prof_mutex);
    for (auto &r : ovm::g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        if (!agg.count(r.label)) order.push_back(r.label);
        auto &e = agg[r.label];
        e.calls++;
        e.ms += ms;
        if (r.side) e.side += ms;
        else if (r.depth > 0) e.nested += ms;
    }
    std::string out;
    for (auto &l : order) {
        char line[256];
        snprintf(line, sizeof line, "%s %llu %.6f %.6f %.6f\n", l.c_str(), (unsigned long long)agg[l].calls, agg[l].ms, agg[l].nested,
                 agg[l].side);
        out += line;
    }
    if (buf && cap) {
        size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return out.size() + 1;
}

// Releases the calling thread's device resources now (mailboxes, pinned upload arena, grind scratch, free pool
// segments); the thread may keep using the library afterwards -- they are created again on demand.
int ovm_hip_thread_shutdown(void) {
    (void)hipDeviceSynchronize();
    // (a resource's release() does not unregister it: the objects live until the thread exits)
    for (ovm::ThreadResource *r : std::vector<ovm::ThreadResource *>(ovm::thread_resources().list)) r->release();
    return 0;
}

uint32_t ovm_hip_abi_version(void) { return 3; }  // 3: ovm_shard_t.abort

int ovm_hip_init(int device, uint32_t *cu_count) {
    OVM_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    OVM_CHECK(hipGetDeviceProperties(&prop, device));
    if (cu_count) *cu_count = (uint32_t)prop.multiProcessorCount;
    if (!ovm::twiddles()) return (int)hipErrorInitializationError;
    return 0;
}

const char *ovm_hip_error_name(int code) {
    switch (code) {
        case OVM_ESHARD_TIMEOUT: return "OVM_ESHARD_TIMEOUT";
        case OVM_ESHARD_PEER_ABORTED: return "OVM_ESHARD_PEER_ABORTED";
        case OVM_ESHARD_POISONED: return "OVM_ESHARD_POISONED";
        default: return hipGetErrorName((hipError_t)code);
    }
}

int ovm_hip_malloc(void **d_ptr, size_t bytes) {
    if (!d_ptr) return (int)hipErrorInvalidValue;
    OVM_CHECK(hipMalloc(d_ptr, bytes ? bytes : 4));
    return 0;
}
int ovm_hip_free(void *d_ptr) {
    OVM_CHECK(hipFree(d_ptr));
    return 0;
}
int ovm_hip_memcpy_h2d(void *d_dst, const void *src, size_t bytes, ovm_stream_t stream) {
    OVM_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    OVM_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int ovm_hip_memcpy_d2h(void *dst, const void *d_src, size_t bytes, ovm_stream_t stream) {
    OVM_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    OVM_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int ovm_hip_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, ovm_stream_t stream) {
    OVM_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
int ovm_hip_memset(void *d_dst, int value, size_t bytes, ovm_stream_t stream) {
    OVM_CHECK(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
    return 0;
}
int ovm_hip_stream_create(ovm_stream_t *stream) {
    hipStream_t s;
    OVM_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (ovm_stream_t)s;
    return 0;
}
int ovm_hip_stream_destroy(ovm_stream_t stream) {
    OVM_CHECK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
int ovm_hip_stream_synchronize(ovm_stream_t stream) {
    OVM_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

}  // extern "C"
