// Host side of the HIP prover device: the three `ProverDevice` stages over resident traces, and
// the Coordinator flow that strings them together with the Fiat-Shamir transcript.
//
// Mirrors (reference): crates/stark-backend/src/prover/mod.rs:104-197 (Coordinator::prove),
// prover/hal.rs:84-138 (TraceCommitter / MultiRapProver / OpeningProver), and the stage drivers of
// the CUDA backend it is a peer of: crates/cuda-backend/src/{stacked_pcs.rs:50-88,
// logup_zerocheck/mod.rs:119-, stacked_reduction.rs, whir.rs:63-420}. The protocol steps and the
// transcript order are those of crates/stark-backend/src/prover/{logup_zerocheck/mod.rs,
// stacked_reduction.rs, whir.rs}; which work runs where (device kernels vs host algebra on a
// handful of extension-field elements) is this backend's design.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <unordered_map>

#include "block_pool.h"
#include "gkr_shard.h"
#include "kernels.h"
#include "shard.h"

namespace ovm {

// ---------------------------------------------------------------------------------------------
// Device memory helpers
// ---------------------------------------------------------------------------------------------
// Device memory comes from the per-(thread, device) pool of block_pool.h: segments kept from the driver, blocks split
// and merged inside them. `ovm_hip_trim` returns the free segments, ovm_hip_memory_stats / ovm_hip_pool_stats report.
// Memory-bounded modes (reference: GpuProverConfig, cuda-backend/src/device.rs:25-30), per
// prover thread (one engine per thread). Defaults keep everything resident: 288 GB of HBM is the normal case.
struct ProverConfig {
    bool cache_rs_code_matrix = true;    // false: a commitment's codeword is dropped once its Merkle tree is built
                                         // and re-encoded when WHIR opens rows of it (cuda-backend/src/whir.rs:404-416)
    bool zerocheck_save_memory = false;  // true: round 0 extends and evaluates a tall trace in slabs of rows
    unsigned round0_slab_log_rows = 20;  // slab size in that mode
    bool gkr_drop_leaves = false;        // true: the leaf layer of the LogUp fraction tree is dropped once the layer above
                                         // it is built and generated again for the last layer's sumcheck
};
inline ProverConfig &prover_config() {
    static thread_local ProverConfig cfg;
    return cfg;
}

// The calling thread's pool for the current device. Pools are heap objects: one whose thread has exited lives on
// until the last of its blocks has come back (BlockPool::orphan).
struct PoolHolder : ThreadResource {
    BlockPool *pools[16] = {};
    void release() override {  // (ovm_hip_thread_shutdown: what is free goes back to the driver, live blocks stay valid)
        for (BlockPool *p : pools)
            if (p) p->trim();
    }
    ~PoolHolder() override {
        if (runtime_exiting()) return;
        for (BlockPool *p : pools)
            if (p && p->orphan()) delete p;
    }
};
inline BlockPool &block_cache() {
    static thread_local PoolHolder holder;
    int dev = 0;
    (void)hipGetDevice(&dev);
    BlockPool *&p = holder.pools[dev & 15];
    if (!p) p = new BlockPool();
    return *p;
}
inline void *pool_get(size_t bytes) {
    try {
        return block_cache().get(bytes);
    } catch (const BlockPool::Error &e) {
        throw OvmFail(e.code, "hipMalloc");
    }
}
inline void pool_put(void *p);

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() {}
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            n = o.n;
            o.p = nullptr;
            o.n = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        p = (T *)pool_get((count ? count : 1) * sizeof(T));
    }
    void release() {
        if (p) pool_put(p);
        p = nullptr;
        n = 0;
    }
};

// Per-thread pinned staging arena. Uploads are bump-allocated from it and issued asynchronously
// (a D2H/H2D copy from pageable memory takes a slow bounce-buffer path and an upload that
// synchronises per call costs ~20 us: with tens of AIRs per proof those add up). The arena is
// rewound whenever the stream is known to be idle past every issued copy (Ctx::sync, Mailbox::wait).
struct PinnedStage : ThreadResource {
    char *p = nullptr;
    char *dp = nullptr;  // the device's address of the same memory (hipHostGetDevicePointer): what the scatter kernel reads
    size_t bytes = 0, used = 0;
    uint64_t bumps = 0;  // allocations ever made: Mailbox::wait rewinds only if none came after its arm()
    void release() override;
    ~PinnedStage() override {
        if (p && !runtime_exiting()) (void)hipHostFree(p);
    }
    void ensure(size_t need) {
        if (need <= bytes) return;
        if (p) (void)hipHostFree(p);
        bytes = std::max<size_t>(need, size_t(16) << 20);
        used = 0;
        void *q = nullptr, *dq = nullptr;
        // (mapped: the scatter kernel of flush_uploads reads the staged bytes straight from this memory)
        if (hipHostMalloc(&q, bytes, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dq, q, 0) != hipSuccess) {
            if (q) (void)hipHostFree(q);
            q = dq = nullptr;
            bytes = 0;
        }
        p = (char *)q;
        dp = (char *)dq;
    }
    const void *device_view(const void *host) const { return dp + ((const char *)host - p); }
    // nullptr when the request does not fit any more: the caller synchronises, rewinds and retries
    void *bump(size_t need) {
        need = (need + 63) & ~size_t(63);
        if (!p) ensure(need);
        if (!p || used + need > bytes) return nullptr;
        void *r = p + used;
        used += need;
        bumps++;
        return r;
    }
    void rewind();
};
inline PinnedStage &pinned_stage() {
    static thread_local PinnedStage st;
    return st;
}

// Small uploads staged in the arena and not yet copied: flush_uploads writes them all with one kernel, in front of
// the next launch on their stream (device_ctx.h: OVM_LAUNCH), of a synchronisation or of a download.
struct UploadJob {
    const void *src;  // in the pinned arena (device-visible)
    void *dst;
    uint64_t bytes;
    uint32_t wg0, gx;
};
static constexpr uint32_t UPLOAD_JOBS_PER_LAUNCH = 24;
static constexpr uint32_t UPLOAD_WG_BYTES = 256 * 16;
static constexpr size_t UPLOAD_DEFER_MAX = size_t(256) << 10;  // larger uploads are copied by the runtime as before
struct UploadJobsArg {
    UploadJob job[UPLOAD_JOBS_PER_LAUNCH];
    uint32_t n;
};
__global__ __launch_bounds__(256) void upload_scatter_kernel(UploadJobsArg a) {
    uint32_t k = 0;
    while (k + 1 < a.n && a.job[k + 1].wg0 <= blockIdx.x) k++;
    const UploadJob j = a.job[k];
    const uint64_t first = (uint64_t)(blockIdx.x - j.wg0) * UPLOAD_WG_BYTES;
    const uint64_t last = first + UPLOAD_WG_BYTES < j.bytes ? first + UPLOAD_WG_BYTES : j.bytes;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(j.src);
    uint4 *d4 = reinterpret_cast<uint4 *>(j.dst);
    const uint64_t i = first / 16 + threadIdx.x;
    if (i * 16 + 16 <= last) {
        d4[i] = s4[i];
    } else if (i * 16 < last) {  // the tail of a buffer whose size is not a multiple of 16
        const uint8_t *s1 = reinterpret_cast<const uint8_t *>(j.src);
        uint8_t *d1 = reinterpret_cast<uint8_t *>(j.dst);
        for (uint64_t b = i * 16; b < last; b++) d1[b] = s1[b];
    }
}
struct PendingUploads {
    std::vector<UploadJob> jobs;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // recorded behind a flush whose launch goes to ANOTHER stream than the uploads' own
    void add(const void *src, void *dst, size_t bytes, hipStream_t s) {
        if (!jobs.empty() && stream != s) flush();  // (one stream per prover thread: this does not happen)
        stream = s;
        jobs.push_back(UploadJob{src, dst, bytes, 0, (uint32_t)((bytes + UPLOAD_WG_BYTES - 1) / UPLOAD_WG_BYTES)});
    }
    // a destination that goes back to the pool before it was written must not be written later
    void forget(void *dst) {
        for (size_t i = 0; i < jobs.size(); i++)
            if (jobs[i].dst == dst) {
                jobs.erase(jobs.begin() + i);
                return;
            }
    }
    void flush() {
        for (size_t at = 0; at < jobs.size(); at += UPLOAD_JOBS_PER_LAUNCH) {
            UploadJobsArg a{};
            a.n = (uint32_t)std::min<size_t>(UPLOAD_JOBS_PER_LAUNCH, jobs.size() - at);
            uint32_t wg = 0;
            for (uint32_t k = 0; k < a.n; k++) {
                a.job[k] = jobs[at + k];
                a.job[k].wg0 = wg;
                wg += a.job[k].gx;
            }
            hipLaunchKernelGGL(upload_scatter_kernel, dim3(wg), dim3(256), 0, stream, a);
        }
        jobs.clear();
    }
};
inline PendingUploads &pending_uploads() {
    static thread_local PendingUploads pu;
    return pu;
}
inline void pool_put(void *p) {
    pending_uploads().forget(p);
    BlockPool::put(p, &block_cache());
}
// `launch_stream`: the stream of the kernel about to be launched (nullptr = the uploads' own). The scatter kernel runs on
// the stream the uploads were staged for; a launch on a different stream (a second Device used from the same thread)
// is ordered behind it through an event -- it used to run with no ordering after the writes of its own arguments
// (ADVICE r2).
void flush_uploads(hipStream_t launch_stream) {
    PendingUploads &pu = pending_uploads();
    if (pu.jobs.empty()) return;
    const hipStream_t own = pu.stream;
    pu.flush();
    if (launch_stream != nullptr && launch_stream != own) {
        if (!pu.done && hipEventCreateWithFlags(&pu.done, hipEventDisableTiming) != hipSuccess) pu.done = nullptr;
        if (pu.done && hipEventRecord(pu.done, own) == hipSuccess) (void)hipStreamWaitEvent(launch_stream, pu.done, 0);
        else (void)hipStreamSynchronize(own);
    }
}
inline void PinnedStage::release() {
    if (!pending_uploads().jobs.empty()) return;  // (staged bytes still to be copied)
    if (p) (void)hipHostFree(p);
    p = dp = nullptr;
    bytes = used = 0;
}
inline void PinnedStage::rewind() {
    if (pending_uploads().jobs.empty()) used = 0;  // (staged bytes that are still to be copied stay where they are)
}

// Host mailbox of the reduction kernels: a host-mapped, coherent page the last workgroup of a
// reduction writes its results and a sequence flag into; the host spins on the flag instead of
// issuing a D2H copy and a stream synchronisation per sumcheck round.
struct Mailbox : ThreadResource {
    static constexpr size_t SLOTS = 65536;  // extension-field result slots (1 MiB page)
    struct Page {
        uint32_t flag;
        uint32_t pad[3];
        EF data[SLOTS];
    };
    Page *h = nullptr, *d = nullptr;
    uint32_t *d_counter = nullptr;
    uint32_t seq = 0;
    uint64_t bumps_at_arm = 0;  // pinned-stage allocations at the time of the last arm()
    void release() override {
        if (h) (void)hipHostFree(h);
        if (d_counter) (void)hipFree(d_counter);
        h = d = nullptr;
        d_counter = nullptr;
    }
    ~Mailbox() override {
        if (!runtime_exiting()) release();
    }
    void init() {
        if (h) return;
        void *hp = nullptr, *dp = nullptr;
        OVM_THROW_IF(hipHostMalloc(&hp, sizeof(Page), hipHostMallocMapped | hipHostMallocCoherent));
        OVM_THROW_IF(hipHostGetDevicePointer(&dp, hp, 0));
        memset(hp, 0, sizeof(Page));
        h = (Page *)hp;
        d = (Page *)dp;
        // 64 bytes of counters, then the 64 accumulators of the atomic finish (device_common.h: RedOut::acc)
        OVM_THROW_IF(hipMalloc((void **)&d_counter, 64 + 64 * 8));
        OVM_THROW_IF(hipMemset(d_counter, 0, 64 + 64 * 8));
    }
    // call once per prove, before the first reduction (a failed launch may have left the counter dirty)
    void reset(hipStream_t s) {
        init();
        OVM_THROW_IF(hipMemsetAsync(d_counter, 0, 64 + 64 * 8, s));
    }
    // Results of this reduction go to slots [slot, slot + K). Several reductions may be in flight on
    // the stream with disjoint slot ranges; they complete in stream order, so waiting for the one
    // armed last makes all of them readable.
    RedOut arm(EF *partials, size_t slot = 0) {
        if (slot + 64 > SLOTS) throw OvmFail((int)hipErrorInvalidValue, "mailbox: slot out of range");
        ++seq;
        bumps_at_arm = pinned_stage().bumps;
        return RedOut{partials, d_counter, d->data + slot, &d->flag, seq, (unsigned long long *)(d_counter + 16)};
    }
    // device address of result slot `slot` (for kernels that finish several reductions, see mle_round_eval_batched)
    EF *slot_ptr(size_t slot) const {
        if (slot + 64 > SLOTS) throw OvmFail((int)hipErrorInvalidValue, "mailbox: slot out of range");
        return d->data + slot;
    }
    // Wait for the reduction armed last and copy out its first n results.
    std::vector<EF> take(size_t n, hipStream_t s) {
        wait(s);
        return read(0, n);
    }
    std::vector<EF> read(size_t slot, size_t n) const {
        if (slot + n > SLOTS) throw OvmFail((int)hipErrorInvalidValue, "mailbox: too many results");
        std::vector<EF> v(n);
        memcpy((void *)v.data(), (const void *)(h->data + slot), n * sizeof(EF));
        return v;
    }
    void wait(hipStream_t s) {
        volatile uint32_t *flag = &h->flag;
        uint64_t spins = 0;
        auto t0 = std::chrono::steady_clock::now();
        while (*flag != seq) {
            if ((++spins & 0xFFFF) == 0) {
                // a faulted kernel never raises the flag: look at the stream from time to time
                hipError_t q = hipStreamQuery(s);
                if (q != hipSuccess && q != hipErrorNotReady) throw OvmFail((int)q, "reduction kernel failed");
                if (q == hipSuccess && *flag != seq &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
                    throw OvmFail((int)hipErrorUnknown, "mailbox: stream idle but no result arrived");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        // Every copy issued before the reduction armed last has completed. A copy staged AFTER that arm() may
        // still be queued behind the kernel: then the arena is left alone (the next synchronisation rewinds it).
        if (pinned_stage().bumps == bumps_at_arm) pinned_stage().rewind();
    }
};
inline Mailbox &mailbox() {
    static thread_local Mailbox boxes[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return boxes[dev & 15];
}

// Small downloads without a runtime copy: a kernel writes the words into a host-mapped, coherent page and raises a
// sequence flag the host spins on -- the 33 `download` calls of a config D proof (roots, final rows, opened rows, final
// polynomials) each cost a copy command + a stream synchronisation, ~41 us of device idle apiece under the profiler
// (VERDICT r2 item 3). Larger transfers keep using the copy engine.
static constexpr size_t DOWNLOAD_PAGE_BYTES = size_t(1) << 20;
static constexpr uint32_t DOWNLOAD_WG_WORDS = 256 * 16;
__global__ __launch_bounds__(256) void download_kernel(const uint32_t *src, uint32_t n_words, uint32_t *page, uint32_t *counter,
                                                       uint32_t *flag, uint32_t seq) {
    const uint32_t first = blockIdx.x * DOWNLOAD_WG_WORDS;
    const uint32_t last = first + DOWNLOAD_WG_WORDS < n_words ? first + DOWNLOAD_WG_WORDS : n_words;
    for (uint32_t i = first + threadIdx.x; i < last; i += 256) st_word_sys(page + i, src[i]);
    release_to_host();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gridDim.x == 1 || atomicAdd(counter, 1u) == gridDim.x - 1) {
            if (gridDim.x > 1) *counter = 0;
            release_to_host();
            st_word_sys(flag, seq);
        }
    }
}
struct DownloadPage : ThreadResource {
    uint32_t *h = nullptr, *d = nullptr;  // [0] = flag, data from word 16 on
    uint32_t *d_counter = nullptr;
    uint32_t seq = 0;
    void release() override {
        if (h) (void)hipHostFree(h);
        if (d_counter) (void)hipFree(d_counter);
        h = d = d_counter = nullptr;
    }
    ~DownloadPage() override {
        if (!runtime_exiting()) release();
    }
    void init() {
        if (h) return;
        void *hp = nullptr, *dp = nullptr;
        OVM_THROW_IF(hipHostMalloc(&hp, DOWNLOAD_PAGE_BYTES + 64, hipHostMallocMapped | hipHostMallocCoherent));
        OVM_THROW_IF(hipHostGetDevicePointer(&dp, hp, 0));
        memset(hp, 0, 64);
        h = (uint32_t *)hp, d = (uint32_t *)dp;
        OVM_THROW_IF(hipMalloc((void **)&d_counter, 64));
        OVM_THROW_IF(hipMemset(d_counter, 0, 64));
    }
    // copies n_words from device memory `src` (4-byte aligned) into out; stream-ordered after what was enqueued
    void fetch(const void *src, void *out, size_t n_words, hipStream_t s) {
        init();
        ++seq;
        const uint32_t grid = (uint32_t)((n_words + DOWNLOAD_WG_WORDS - 1) / DOWNLOAD_WG_WORDS);
        OVM_LAUNCH(download_kernel, dim3(grid), dim3(256), 0, s, (const uint32_t *)src, (uint32_t)n_words, d + 16, d_counter, d, seq);
        OVM_THROW_IF(hipGetLastError());
        volatile uint32_t *flag = h;
        uint64_t spins = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (*flag != seq) {
            if ((++spins & 0xFFFF) == 0) {  // a faulted kernel never raises the flag
                const hipError_t q = hipStreamQuery(s);
                if (q != hipSuccess && q != hipErrorNotReady) throw OvmFail((int)q, "download kernel failed");
                if (q == hipSuccess && *flag != seq && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
                    throw OvmFail((int)hipErrorUnknown, "download: stream idle but no data arrived");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        memcpy(out, (const void *)(h + 16), n_words * 4);
    }
};
inline DownloadPage &download_page() {
    static thread_local DownloadPage pages[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return pages[dev & 15];
}

// The library's own second stream (lowest priority) for work that needs no challenge: the device spends the first
// layers of the GKR sumcheck waiting for the host (a handful of pairs per launch, a mailbox round trip per sub-round),
// and round 0's coset LDE -- a function of the traces alone -- runs beside them (prove_rap_constraints: EarlyLde).
// fork: recorded on the caller's stream, waited for here; join: recorded here, waited for on the caller's stream.
struct SideStream : ThreadResource {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    void release() override {
        if (join) (void)hipEventDestroy(join);
        if (fork) (void)hipEventDestroy(fork);
        if (s) (void)hipStreamDestroy(s);
        s = nullptr, fork = join = nullptr;
    }
    ~SideStream() override {
        if (!runtime_exiting()) release();
    }
    void init() {
        if (s) return;
        // (a CU-masked stream -- hipExtStreamCreateWithCUMask leaving four CUs per XCD to the caller's stream -- was
        // tried: the two streams then do not overlap at all, 79.1 ms against 74.6 ms)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess ||
            hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            release();  // (all or nothing: a half-made set would pass the `if (s)` above next time)
            throw OvmFail((int)hipErrorUnknown, "side stream");
        }
    }
};
inline SideStream &side_stream() {
    static thread_local SideStream streams[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return streams[dev & 15];
}

struct Ctx {
    hipStream_t stream;
    void sync() const {
        flush_uploads(stream);
        OVM_THROW_IF(hipStreamSynchronize(stream));
        pinned_stage().rewind();
    }
    // Asynchronous: the host vector is copied into the staging arena before returning, the device
    // copy is ordered on the stream before every later launch.
    template <class T>
    DevBuf<T> upload(const std::vector<T> &v) const {
        DevBuf<T> b(v.size());
        if (!v.empty()) {
            const size_t bytes = v.size() * sizeof(T);
            PinnedStage &ps = pinned_stage();
            void *stage = ps.bump(bytes);
            if (!stage) {
                sync();
                ps.ensure(bytes);
                stage = ps.bump(bytes);
            }
            if (stage && bytes <= UPLOAD_DEFER_MAX) {
                memcpy(stage, v.data(), bytes);
                pending_uploads().add(ps.device_view(stage), b.p, bytes, stream);  // written by the next flush_uploads
            } else if (stage) {
                memcpy(stage, v.data(), bytes);
                OVM_THROW_IF(hipMemcpyAsync(b.p, stage, bytes, hipMemcpyHostToDevice, stream));
            } else {
                OVM_THROW_IF(hipMemcpyAsync(b.p, v.data(), bytes, hipMemcpyHostToDevice, stream));
                sync();
            }
        }
        return b;
    }
    template <class T>
    std::vector<T> download(const T *d, size_t n) const {
        std::vector<T> v(n);
        if (n) {
            const size_t bytes = n * sizeof(T);
            static_assert(sizeof(T) % 4 == 0, "downloads are word-granular");
            if (bytes <= DOWNLOAD_PAGE_BYTES && tuning().download_kernel) {
                download_page().fetch(d, v.data(), bytes / 4, stream);
                pinned_stage().rewind();  // (the stream has drained up to the copy kernel: staged uploads were consumed)
                return v;
            }
            flush_uploads(stream);
            PinnedStage &ps = pinned_stage();
            void *stage = ps.bump(bytes);
            if (!stage) {
                sync();
                ps.ensure(bytes);
                stage = ps.bump(bytes);
            }
            if (stage) {
                OVM_THROW_IF(hipMemcpyAsync(stage, d, bytes, hipMemcpyDeviceToHost, stream));
                OVM_THROW_IF(hipStreamSynchronize(stream));
                memcpy(v.data(), stage, bytes);
                ps.rewind();
            } else {
                OVM_THROW_IF(hipMemcpyAsync(v.data(), d, bytes, hipMemcpyDeviceToHost, stream));
                sync();
            }
        }
        return v;
    }
};

// Stacking of many traces in ONE launch (a 36-AIR segment was 36 device copies plus a launch per column of every
// short trace: 3.4 ms for 1.6 GB). Job j owns workgroups [wg0, wg0 + gx): dst[i * stride] = src[i] for i < n;
// stride 1 (a whole trace, or any column run that is contiguous in the stacked matrix) moves 16 bytes per lane.
struct StackJob {
    const Fp *src;
    Fp *dst;
    uint64_t n, stride;
    uint32_t wg0, gx;
};
static constexpr uint32_t STACK_WG_ELEMS = 256 * 16;
__global__ __launch_bounds__(256) void stack_batched_kernel(const StackJob *jobs, uint32_t n_jobs) {
    uint32_t lo = 0, hi = n_jobs - 1;
    while (lo < hi) {  // the job of this workgroup
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (jobs[mid].wg0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const StackJob j = jobs[lo];
    const uint64_t first = (uint64_t)(blockIdx.x - j.wg0) * STACK_WG_ELEMS;
    const uint64_t last = first + STACK_WG_ELEMS < j.n ? first + STACK_WG_ELEMS : j.n;
    if (j.stride == 1 && (((uintptr_t)j.src | (uintptr_t)j.dst) & 15) == 0 && (j.n & 3) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(j.src);
        uint4 *d4 = reinterpret_cast<uint4 *>(j.dst);
        for (uint64_t i = first / 4 + threadIdx.x; i < last / 4; i += 256) d4[i] = s4[i];
    } else {
        for (uint64_t i = first + threadIdx.x; i < last; i += 256) j.dst[i * j.stride] = j.src[i];
    }
}

// eq'[x] = eq[x]*a ; krot[x] = b*eq[x] + c*(eq[prev x] - eq[x])
__global__ void stacked_tables_kernel(const EF *eq_in, uint64_t n, EF a, EF b, EF c, EF *eq_out, EF *krot_out) {
    uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    EF e = eq_in[x], ep = eq_in[x == 0 ? n - 1 : x - 1];
    eq_out[x] = e * a;
    krot_out[x] = b * e + c * (ep - e);
}

// tab[q * 2^nb + x] = scale_q * prod_{j < nb} (x_j ? zp[q][b0 + j] : 1 - zp[q][b0 + j])
__global__ void whir_eq_tables_kernel(const EF *zpows, uint32_t dim, uint32_t b0, uint32_t nb, const EF *scale,
                                      uint32_t n_q, EF *tab) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((uint64_t)n_q << nb)) return;
    uint32_t q = (uint32_t)(i >> nb), x = (uint32_t)(i & ((1u << nb) - 1));
    EF acc = scale ? scale[q] : EF::one();
    for (uint32_t j = 0; j < nb; j++) {
        EF z = zpows[(uint64_t)q * dim + b0 + j];
        acc = acc * ((x >> j & 1) ? z : EF::one() - z);
    }
    tab[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// Inputs / PCS data
// ---------------------------------------------------------------------------------------------
struct PcsData;
struct TraceIn {
    uint32_t air_idx;
    const Fp *d_trace;  // common main, column-major, width from the key
    size_t height;
    std::vector<Fp> public_values;
    std::vector<const PcsData *> cached;  // committed cached main partitions, in partition order (owner only)
    std::vector<p2::Digest> cached_roots;  // their commitments (every rank, after the exchange of a sharded proof)
    uint32_t owner = 0;
};

struct PcsData {
    Layout layout;
    unsigned log_height = 0, log_rows = 0, k = 0;
    DevBuf<Fp> stacked, msg, codeword;
    const Fp *stacked_view = nullptr;  // the stacked matrix: `stacked`, or the caller's trace when that IS the matrix
    // single-matrix commitments (preprocessed / cached main): the committed trace itself
    const Fp *mat_ptr = nullptr;
    size_t mat_w = 0, mat_h = 0;
    DevBuf<Fp> own_mat;  // backing store when the commitment owns the trace (preprocessed, from the key blob)
    DevBuf<p2::Digest> layers;
    p2::Digest root;
    // A tree built by dist_n ranks together (sharded proving: the common-main commitment): `layers` is THIS rank's
    // share -- the tree over the query groups g with g mod dist_n == dist_r, compacted -- and `top` (host) the
    // layers of dist_n, dist_n / 2, ..., 1 entries above the shares' roots, identical on every rank.
    uint32_t dist_n = 1, dist_r = 0;
    bool dist_member = true;  // false: this rank holds no share (ranks beyond the largest power of two)
    std::vector<p2::Digest> top;
    size_t query_stride() const { return size_t(1) << (log_rows - k); }
};

static inline uint64_t bitrev64(uint64_t x, unsigned bits) {
    uint64_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1ull) << (bits - 1 - i);
    return r;
}

// Physical position of entry y of stored Merkle layer j (see merkle.hip).
static inline uint64_t layer_entry(uint64_t qs, unsigned lq, unsigned j, uint64_t y) {
    uint64_t off = 0;
    for (unsigned m = 0; m < j; m++) off += qs >> m;
    return off + bitrev64(y, lq - j);
}

// Host time of named pieces of the prove call, summed and printed with the OVM_TRACE marks (where does a stage's HOST share
// go: a many-AIR proof spends a quarter of its wall time with the device idle).
struct HostTime {
    struct Slot {
        const char *name;
        double ms;
        uint64_t calls;
    };
    static std::vector<Slot> &slots() {
        static thread_local std::vector<Slot> s;
        return s;
    }
    const char *name;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit HostTime(const char *n) : name(n), on(tuning().trace) {
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~HostTime() {
        if (!on) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (auto &s : slots())
            if (s.name == name) {
                s.ms += ms, s.calls++;
                return;
            }
        slots().push_back({name, ms, 1});
    }
    static void report() {
        for (auto &s : slots()) fprintf(stderr, "[ovm] host %-32s %9.3f ms over %llu calls\n", s.name, s.ms, (unsigned long long)s.calls);
        slots().clear();
    }
};

struct WallTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    WallTrace() : on(tuning().trace), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ovm] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Proving key on the device: the parsed key plus, per AIR, the committed preprocessed trace
// (DeviceStarkProvingKey::preprocessed_data, prover/types.rs:43-50); null where the AIR has none.
struct DeviceKey {
    ProvingKey pk;
    std::vector<std::unique_ptr<PcsData>> pre;
};

struct Prover {
    WallTrace wt;
    const DeviceKey &dk;
    const ProvingKey &pk;
    Ctx cx;
    Mailbox &mb;
    // test hook: take the GKR fallback path (three evaluation points from the device) on every sub-round
    const bool force_three_points = tuning().gkr_three_points;
    Transcript ts;
    std::vector<TraceIn> traces;  // sorted
    // Sharded proving: owner[t] = the rank holding sorted trace t. Per-trace device work
    // runs on the owner only; its few results are all-gathered so that every rank runs the same transcript.
    ShardComm sh;
    std::vector<uint32_t> owner;
    bool mine(size_t t) const { return owner[t] == sh.rank; }
    ByteWriter out;
    std::vector<size_t> section_offsets;

    Prover(const DeviceKey &dk_, hipStream_t s) : dk(dk_), pk(dk_.pk), cx{s}, mb(mailbox()) {
        ts.stream = s;
        mb.reset(s);
    }

    const Params &P() const { return pk.params; }
    const AirKey &air(size_t t) const { return pk.airs[traces[t].air_idx]; }
    const PcsData *pre_of(size_t t) const {
        const uint32_t a = traces[t].air_idx;
        return a < dk.pre.size() ? dk.pre[a].get() : nullptr;
    }

    // -----------------------------------------------------------------------------------------
    // Stage 1: commit (TraceCommitter::commit)
    // -----------------------------------------------------------------------------------------
    struct MatIn {
        const Fp *ptr;
        size_t w, h;
    };
    PcsData commit() {
        std::vector<MatIn> mats;
        for (size_t t = 0; t < traces.size(); t++) mats.push_back({traces[t].d_trace, air(t).common_main_width, traces[t].height});
        return commit_mats(mats);
    }
    // stacked_commit (stacked_pcs.rs:103-140) of column-major device matrices sorted by height (descending)
    PcsData commit_mats(const std::vector<MatIn> &mats, bool distributed_tree = false) {
        const unsigned l_skip = P().l_skip, lh = l_skip + P().n_stack;
        PcsData d;
        std::vector<std::pair<size_t, unsigned>> meta;
        size_t total_cells = 0;
        for (auto &m : mats) {
            meta.push_back({m.w, log2_strict(m.h)});
            total_cells += std::max(m.h, size_t(1) << l_skip) * m.w;
        }
        if (mats.size() == 1) {
            d.mat_ptr = mats[0].ptr;
            d.mat_w = mats[0].w;
            d.mat_h = mats[0].h;
        }
        d.layout = Layout(l_skip, lh, meta);
        const size_t H = size_t(1) << lh, width = (total_cells + H - 1) / H;
        if (width != d.layout.width) throw OvmFail(1, "stacked width mismatch");
        d.log_height = lh;
        ProfScope *ps_stack = new ProfScope("commit.stack", cx.stream);
        // Column-major traces stack by concatenation, so ONE trace that fills the matrix exactly already is
        // the stacked matrix (config D: 100 columns of 2^22 = 25 columns of 2^24): no copy, the trace is
        // only ever read. Otherwise: one device copy per trace into a zero-padded buffer.
        const bool alias = mats.size() == 1 && mats[0].h >= (size_t(1) << l_skip) && mats[0].h * mats[0].w == width * H;
        if (alias) {
            d.stacked_view = mats[0].ptr;
        } else {
            d.stacked.alloc(width * H);
            d.stacked_view = d.stacked.p;
            OVM_THROW_IF(hipMemsetAsync(d.stacked.p, 0, width * H * sizeof(Fp), cx.stream));
        }
        std::vector<StackJob> jobs;
        auto add_job = [&](const Fp *src, Fp *dst, uint64_t n, uint64_t stride) {
            if (!n) return;
            StackJob j{src, dst, n, stride, jobs.empty() ? 0u : jobs.back().wg0 + jobs.back().gx, (uint32_t)ceil_div(n, (uint64_t)STACK_WG_ELEMS)};
            jobs.push_back(j);
        };
        for (size_t t = 0; t < mats.size() && !alias; t++) {
            size_t w = mats[t].w, h = mats[t].h;
            if (w == 0) continue;
            const Slice &s0 = d.layout.cols[d.layout.mat_starts[t]].slice;
            if (s0.log_height >= l_skip) {
                // the columns of one trace occupy one contiguous range of the stacked buffer
                add_job(mats[t].ptr, d.stacked.p + s0.col_idx * H + s0.row_idx, (uint64_t)w * h, 1);
            } else {
                const size_t stride = s0.stride(l_skip);
                for (size_t j = 0; j < w; j++) {
                    const Slice &sl = d.layout.cols[d.layout.mat_starts[t] + j].slice;
                    add_job(mats[t].ptr + j * h, d.stacked.p + sl.col_idx * H + sl.row_idx, (uint64_t)h, (uint64_t)stride);
                }
            }
        }
        if (!jobs.empty()) {
            auto d_jobs = cx.upload(jobs);
            OVM_LAUNCH(stack_batched_kernel, dim3(jobs.back().wg0 + jobs.back().gx), dim3(256), 0, cx.stream, d_jobs.p,
                               (uint32_t)jobs.size());
            OVM_THROW_IF(hipGetLastError());
        }
        delete ps_stack;
        d.msg.alloc(width * H);
        OVM_THROW_IF(ovm_hip_eval_to_coeff((const uint32_t *)d.stacked_view, (uint32_t *)d.msg.p, width, lh, l_skip, cx.stream));
        d.log_rows = lh + P().log_blowup;
        d.k = P().k_whir;
        d.codeword.alloc(width << d.log_rows);
        OVM_THROW_IF(ovm_hip_ntt_dif((const uint32_t *)d.msg.p, (uint32_t *)d.codeword.p, width, d.log_rows, lh, cx.stream));
        build_tree(d, (const uint32_t *)d.codeword.p, width, false, distributed_tree);
        if (!prover_config().cache_rs_code_matrix) d.codeword.release();  // re-encoded from `msg` when rows are opened
        return d;
    }

    // distributed (sharded proving, and only for data every rank holds): the ranks hash a share of the query groups
    // each and exchange the shares' roots; rank r < dist_n takes the groups g with g mod dist_n == r. In the
    // bit-reversed layer order of merkle.hip entry p of the layer with dist_n entries covers exactly those groups, so
    // a share is a tree of its own and the layers above are dist_n - 1 compressions on the host.
    void build_tree(PcsData &d, const uint32_t *matrix, size_t width, bool ext, bool distributed = false) {
        uint32_t n = 1;
        if (distributed && !sh.single()) {
            while (n * 2 <= sh.world && n * 2 <= (1u << (d.log_rows - d.k))) n *= 2;
        }
        if (n == 1) {
            DevBuf<uint32_t> scratch(std::max<uint64_t>(1, ovm_hip_merkle_scratch_words(d.log_rows, d.k)));
            d.layers.alloc(ovm_hip_merkle_layers_words(d.log_rows, d.k) / 8);
            if (ext)
                OVM_THROW_IF(ovm_hip_merkle_tree_ext(matrix, width, d.log_rows, d.k, scratch.p, (uint32_t *)d.layers.p, cx.stream));
            else
                OVM_THROW_IF(ovm_hip_merkle_tree(matrix, width, d.log_rows, d.k, scratch.p, (uint32_t *)d.layers.p, cx.stream));
            d.root = cx.download(d.layers.p + d.layers.n - 1, 1)[0];
            return;
        }
        d.dist_n = n;
        d.dist_member = sh.rank < n;
        d.dist_r = d.dist_member ? sh.rank : 0;
        const unsigned log_share = d.log_rows - log2_strict(n);
        p2::Digest mine{};
        if (d.dist_member) {
            DevBuf<uint32_t> scratch(std::max<uint64_t>(1, ovm_hip_merkle_scratch_words(log_share, d.k)));
            d.layers.alloc(ovm_hip_merkle_layers_words(log_share, d.k) / 8);
            OVM_THROW_IF(merkle_tree_share(matrix, width, log_share, d.k, scratch.p, (uint32_t *)d.layers.p, cx.stream, ext, n, d.dist_r));
            mine = cx.download(d.layers.p + d.layers.n - 1, 1)[0];
        }
        std::vector<p2::Digest> all(sh.world);
        sh.allgather(&mine, all.data(), sizeof(p2::Digest));
        d.top.assign(all.begin(), all.begin() + n);
        for (size_t lo = 0, size = n; size > 1; lo += size, size /= 2)
            for (size_t o = 0; o < size / 2; o++) d.top.push_back(p2::compress(d.top[lo + o], d.top[lo + o + size / 2]));
        d.root = d.top.back();
    }

    // -----------------------------------------------------------------------------------------
    // Per-trace device view
    // -----------------------------------------------------------------------------------------
    struct TraceView {
        ColumnMap cm;
        unsigned log_height = 0, log_lifted = 0, n_lift = 0;
        int n = 0;
        DevBuf<Fp> sels;
        DevBuf<ColDesc> cols;
        DevBuf<int32_t> twins;  // per flat column: index of its rotated view / -2 for the views / -1 (fold_ple)
        // folded matrices (EF, column-major, all flat columns), ping-pong
        DevBuf<EF> mats_a, mats_b;
        bool mats_in_a = true;
        unsigned cur_log_h = 0;
        std::vector<EF> final_row;  // once height 1
    };

    struct ProgDev {
        Program prog;
        DevBuf<Instr> code;
        DevBuf<EF> weights;
        // round-0 and leaf programs also carry their staged form (program.h: stage_program), one
        // device buffer: [blocks | accs | code | loads]
        DevBuf<uint32_t> st_words;
        DevStaged staged;
        DevProgram dev() const { return DevProgram{code.p, (uint32_t)prog.code.size(), prog.num_slots, weights.p}; }
    };
    // staged: also build the staged form, laid out for the round-0 launch on 2^log_lifted rows with
    // n_out outputs (log_lifted 0: one row per lane, the leaves kernel). The plain code is then only
    // uploaded if staging does not fit.
    ProgDev make_prog(Program p, const std::vector<EF> &weights, bool staged = false, uint32_t log_lifted = 0,
                      uint32_t n_out = 1) {
        ProgDev d;
        d.weights = cx.upload(weights);
        HostTime ht_all("make_prog");
        const StagedPlan plan = staged ? round0_staged_plan(log_lifted, n_out, p.num_slots, (uint32_t)p.code.size()) : StagedPlan();
        if (plan.rows == 0) {
            d.code = cx.upload(p.code);
            d.prog = std::move(p);
            return d;
        }
        StagedLayout layout;
        layout.stage_rows = plan.stage_rows;
        layout.acc_rows = plan.acc_rows;
        StagedProgram sp;
        {
            HostTime ht("make_prog.stage_program");
            sp = stage_program(p, weights, layout, tuning().staged_forward);
            sp.scale_rows(plan.rows);
        }
        static_assert(sizeof(StagedBlock) == 32 && sizeof(StagedAcc) == 32 && sizeof(StagedInstr) == 16, "staged layout");
        const size_t w_blocks = sp.blocks.size() * 8, w_accs = sp.accs.size() * 8, w_code = sp.code.size() * 4;
        std::vector<uint32_t> words(w_blocks + w_accs + w_code + sp.loads.size());
        memcpy(words.data(), sp.blocks.data(), w_blocks * 4);
        memcpy(words.data() + w_blocks, sp.accs.data(), w_accs * 4);
        memcpy(words.data() + w_blocks + w_accs, sp.code.data(), w_code * 4);
        memcpy(words.data() + w_blocks + w_accs + w_code, sp.loads.data(), sp.loads.size() * 4);
        d.st_words = cx.upload(words);
        const uint32_t *base = d.st_words.p;
        d.staged = DevStaged{(const StagedBlock *)base, base + w_blocks + w_accs + w_code,
                             (const StagedInstr *)(base + w_blocks + w_accs), (const StagedAcc *)(base + w_blocks),
                             (uint32_t)sp.blocks.size(), sp.num_rows, plan.rows, layout.stage_rows};
        d.prog = std::move(p);
        return d;
    }

    // A grouped program (disjoint sets of accumulation roots, the same groups the multilinear rounds use)
    // with every group staged on its own: round 0 of a LONG program on a SHORT trace runs the groups as
    // extra workgroups of one job instead of walking thousands of instructions in a few lanes.
    struct StagedGroups {
        DevBuf<uint32_t> words;
        DevBuf<DevStaged> descs;
        std::vector<DevStaged> host;
        uint32_t rows = 0, max_num_rows = 0;
    };
    // want_groups <= the program's groups: consecutive groups are merged (their codes are self-contained, so
    // running one after the other in a lane is the sum of their accumulations)
    std::unique_ptr<StagedGroups> make_staged_groups(const GroupedProgram &gp, const std::vector<EF> &weights,
                                                     uint32_t log_lifted, uint32_t n_out, size_t want_groups) {
        const StagedPlan plan = round0_staged_plan(log_lifted, n_out, gp.max_slots);
        const size_t src_groups = gp.off.size() - 1;
        const size_t n_groups = std::min(src_groups, want_groups);
        if (plan.rows == 0 || n_groups < 2) return nullptr;
        StagedLayout layout;
        layout.stage_rows = plan.stage_rows;
        layout.acc_rows = plan.acc_rows;
        auto sg = std::make_unique<StagedGroups>();
        sg->rows = plan.rows;
        std::vector<uint32_t> words;
        struct Off {
            size_t blocks, accs, code, loads, n_blocks, num_rows;
        };
        std::vector<Off> offs;
        for (size_t k = 0; k < n_groups; k++) {
            Program pg;
            pg.code.assign(gp.code.begin() + gp.off[src_groups * k / n_groups], gp.code.begin() + gp.off[src_groups * (k + 1) / n_groups]);
            pg.num_slots = gp.max_slots;
            StagedProgram sp = stage_program(pg, weights, layout, tuning().staged_forward);
            sp.scale_rows(plan.rows);
            while (words.size() % 8) words.push_back(0);  // 32-byte alignment of the block / entry arrays
            const size_t w_blocks = sp.blocks.size() * 8, w_accs = sp.accs.size() * 8, w_code = sp.code.size() * 4;
            const size_t base = words.size();
            words.resize(base + w_blocks + w_accs + w_code + sp.loads.size());
            memcpy(words.data() + base, sp.blocks.data(), w_blocks * 4);
            memcpy(words.data() + base + w_blocks, sp.accs.data(), w_accs * 4);
            memcpy(words.data() + base + w_blocks + w_accs, sp.code.data(), w_code * 4);
            memcpy(words.data() + base + w_blocks + w_accs + w_code, sp.loads.data(), sp.loads.size() * 4);
            offs.push_back({base, base + w_blocks, base + w_blocks + w_accs, base + w_blocks + w_accs + w_code, sp.blocks.size(),
                            sp.num_rows});
            sg->max_num_rows = std::max(sg->max_num_rows, sp.num_rows);
        }
        sg->words = cx.upload(words);
        const uint32_t *b = sg->words.p;
        for (auto &o : offs)
            sg->host.push_back(DevStaged{(const StagedBlock *)(b + o.blocks), b + o.loads, (const StagedInstr *)(b + o.code),
                                         (const StagedAcc *)(b + o.accs), (uint32_t)o.n_blocks, (uint32_t)o.num_rows, plan.rows,
                                         layout.stage_rows});
        sg->descs = cx.upload(sg->host);
        return sg;
    }

    // host evaluation of a program on one row (tail rounds)
    static void host_eval(const Program &p, const std::vector<EF> &weights, const std::vector<EF> &row, EF add_const,
                          EF out[2]) {
        std::vector<EF> slots(p.num_slots);
        auto fetch = [&](uint32_t col) { return row[col]; };
        out[0] = EF::zero();
        out[1] = add_const;
        run_program_acc<EF>(p.code.data(), (uint32_t)p.code.size(), weights.data(), slots.data(), fetch, out);
    }

    // The register programs of every trace (they depend on the key and the public values only, not on
    // any challenge): in the one-call prove they are compiled on a helper thread while the device
    // commits the traces -- ~7 ms of host work for a 46-AIR segment that the device used to wait for.
    // (splitting the roots of the short rounds four times finer -- 64 groups -- was measured slower:
    // batched kernel 174 -> 210 ms over the 21-segment replay sample; per-workgroup overhead dominates)
    static constexpr uint32_t MLE_GROUPS = 16;
    // long programs are split finer: a lane's walk through its group is what a short round takes, and a
    // 10 K-instruction program in 16 groups is 600 instructions per lane (OVM_MLE_GROUPS_MAX caps it)
    static constexpr uint32_t MLE_GROUP_INSTRS = 128, MLE_GROUPS_MAX = 128;
    static uint32_t mle_groups_for(size_t n_instr) {
        const uint32_t cap = tuning().mle_groups_max;
        return (uint32_t)std::min<size_t>(std::max<size_t>(MLE_GROUPS, n_instr / MLE_GROUP_INSTRS), std::max(cap, MLE_GROUPS));
    }
    struct TracePrograms {
        Program zc, lg, leaves;
        GroupedProgram zc_g, lg_g;
    };
    std::future<std::vector<TracePrograms>> programs_future;
    std::vector<TracePrograms> compile_all_programs() const {
        std::vector<TracePrograms> out(traces.size());
        for (size_t t = 0; t < traces.size(); t++) {
            if (!mine(t)) continue;
            const AirKey &a = air(t);
            const ColumnMap cm = column_map_of(a);
            const std::vector<Fp> &pv = traces[t].public_values;
            const bool tall = traces[t].height > (size_t(1) << P().l_skip);  // has multilinear rounds
            {
                const std::vector<Root> roots = roots_zerocheck(a);
                out[t].zc = compile_program(a, cm, pv, roots, 0);
                if (tall) out[t].zc_g = compile_grouped(a, cm, pv, roots, 0, mle_groups_for(out[t].zc.code.size()),
                                                        mle_slot_target(pk.params.max_constraint_degree));
            }
            if (!a.interactions.empty()) {
                const std::vector<Root> roots = roots_logup(a);
                out[t].lg = compile_program(a, cm, pv, roots, 0);
                if (tall) out[t].lg_g = compile_grouped(a, cm, pv, roots, 0, mle_groups_for(out[t].lg.code.size()),
                                                        mle_slot_target(pk.params.max_constraint_degree));
                out[t].leaves = compile_program(a, cm, pv, roots_leaves(a), 0);
            }
        }
        return out;
    }

    // Sharded proving: vecs[i] was computed by rank owner_of(i) only; afterwards every rank holds all of them.
    template <class OwnerOf>
    void share_vectors(std::vector<std::vector<EF>> &vecs, OwnerOf owner_of) {
        if (sh.single()) return;
        Packer pk_;
        for (size_t i = 0; i < vecs.size(); i++)
            if (owner_of(i) == sh.rank) {
                pk_.put<uint64_t>(i);
                pk_.put_vec(vecs[i]);
            }
        const auto all = sh.allgatherv(pk_.buf);
        for (uint32_t r = 0; r < sh.world; r++) {
            if (r == sh.rank) continue;
            Unpacker u(all[r]);
            while (!u.done()) {
                const uint64_t i = u.get<uint64_t>();
                if (i >= vecs.size() || owner_of(i) != r) throw OvmFail(1, "shard exchange: unexpected vector");
                vecs[i] = u.get_vec<EF>();
            }
        }
    }

    // -----------------------------------------------------------------------------------------
    // The device side of the LogUp-GKR driver (gkr_shard.h): this rank's window of every tree layer,
    // of the folded tables and of the eq heap. With one rank and no padding the windows are the
    // whole arrays and the launches are the ones a single-device prover makes.
    // -----------------------------------------------------------------------------------------
    static constexpr uint64_t PLANE_SKEW = 1088;  // words: 4 KiB + 256 B
    struct GkrDevice : GkrDeviceOps {
        Prover &pr;
        const GkrPlan &plan;
        const uint32_t me;
        const unsigned TR;
        const bool p_base;  // big trees: only limb 0 of the leaf numerators exists
        std::vector<uint64_t> toff, fb;  // per layer: plane offset of my window, global index of its element 0
        uint64_t tree_stride = 0, leaf_stride = 0, wcap = 0;
        // The layers above the leaves share one buffer (a window per layer); the leaf layer has its own: with
        // base-field numerators it is 5 planes, not 8, and in the memory-bounded mode it is dropped once the layer
        // above exists and generated again for the last layer's sumcheck (make_leaves).
        DevBuf<uint32_t> tree_buf, leaf_buf, rec_a, rec_b, eq_buf;
        // split eq tables of the big layers (begin_layer): grown on demand, a few hundred KiB
        DevBuf<uint32_t> eq_lo_buf, eq_hi_buf;
        uint64_t eq_lo_stride = 0, eq_hi_stride = 0;
        EqHeapPlan hp_lo, hp_hi;
        unsigned split_m = 0;  // 0: this layer reads its weights from the whole heap
        DevBuf<EF> partials;
        EqHeapPlan hp;
        uint64_t eq_stride = 0;
        std::vector<uint64_t> rbase;  // rbase[sr]: global record index of element 0 of table(sr)'s buffer (sr >= 1)
        std::vector<URange> Rchain;   // Rchain[s] = my records of table(s) of the current layer
        // Two sub-rounds per pass (stages.hip: gkr_pass_kernel). Not every table(s) is written any more:
        std::vector<EF> point;        // the current layer's eq point
        unsigned mat_sr = 0;          // the latest table of the current layer that exists on the device (0 = the tree layer)
        std::vector<uint8_t> tbuf;    // tbuf[s]: which of rec_a / rec_b holds table(s)
        std::vector<EF> r_hist;       // r_hist[s] = the challenge that folds table(s) into table(s + 1)
        struct Lookahead {            // what the pass at sub-round sr - 1 left for sub-round sr
            bool valid = false;
            unsigned sr = 0;
            URange pairs;
            EF v[3][2];               // S(X; r) at r = 0, 1, 2 (first index) and X = 1, 2
            unsigned n_src = 0;       // source pairs of table(sr - 1) dumped for sub-round sr's unpaired edge records
            uint64_t src_pair[2] = {0, 0};
            GkrRec src[2][2];
        } la;
        uint64_t la_hits = 0, la_passes = 0, plain_passes = 0;
        static constexpr size_t LA_DUMP_SLOT = 1024;  // mailbox slots of the dumped source records (sums: 0.., edges: 8..)
        bool drop_leaves = false;             // ProverConfig::gkr_drop_leaves (and a tree of at least two layers)
        std::function<void()> make_leaves;    // enqueues the leaf kernels into leaves()

        static uint64_t up64(uint64_t x) { return (x + 63) & ~uint64_t(63); }
        GkrDevice(Prover &pr_, const GkrPlan &pl, bool p_base_)
            : pr(pr_), plan(pl), me(pr_.sh.rank), TR(pl.TR), p_base(p_base_), toff(pl.TR + 1, 0), fb(pl.TR + 1, 0) {
            uint64_t cur = 0;
            for (unsigned j = 0; j < TR; j++) {
                const URange &r = plan.F[j][me];
                toff[j] = cur;
                if (r.empty()) continue;
                fb[j] = r.lo & ~uint64_t(63);
                cur += up64(r.hi - fb[j]);
            }
            // (+ PLANE_SKEW words: limb planes a power of two apart put the 8-20 streams of a wave on the same
            // HBM channels)
            tree_stride = std::max<uint64_t>(cur, 64) + PLANE_SKEW;
            tree_buf.alloc(8 * tree_stride);
            const URange &lr = plan.F[TR][me];
            if (!lr.empty()) fb[TR] = lr.lo & ~uint64_t(63);
            leaf_stride = std::max<uint64_t>(lr.empty() ? 0 : up64(lr.hi - fb[TR]), 64) + PLANE_SKEW;
            alloc_leaves();
            drop_leaves = prover_config().gkr_drop_leaves && TR >= 3;
            partials.alloc(RED_BLOCKS * 8);  // (a pass with lookahead leaves 2 + 6 sums per workgroup)
            // (memory-bounded mode: the sumcheck buffers come when the build is over, sized for the layers above the
            // leaves, and again -- with the tree gone -- for the leaf layer: begin_layer)
            if (!drop_leaves) alloc_round_buffers(TR >= 1 ? TR - 1 : 0);
        }
        // the folded tables and the eq heap of the layer sumchecks up to `last_round`
        void alloc_round_buffers(unsigned last_round) {
            rec_a.release(), rec_b.release(), eq_buf.release();
            // folded tables: table(1) holds at most half of the layer's records (+ alignment slack), 2 fractions each
            wcap = up64(plan.F[std::min(last_round + 1, TR)][me].size() / 2 + 128) + PLANE_SKEW;
            rec_a.alloc(8 * wcap);
            rec_b.alloc(8 * wcap);
            uint64_t eq_total = 64;
            // (the layers that take split tables -- begin_layer -- never touch this buffer: at config D it shrinks from
            // 1.1 GB to a megabyte)
            const bool split = tuning().gkr_split_eq;
            for (unsigned round = 1; round <= last_round && round < TR; round++)
                if (!split || round < tuning().gkr_split_min_round) eq_total = std::max(eq_total, heap_plan_for(round).total);
            eq_stride = up64(eq_total) + PLANE_SKEW;
            eq_buf.alloc(4 * eq_stride);
        }
        void alloc_leaves() { leaf_buf.alloc((p_base ? 5 : 8) * leaf_stride); }
        FracPlanes tree() const { return FracPlanes::make(tree_buf.p, tree_stride); }
        // element 0 = global leaf fb[TR]; base-field numerators: ONE numerator plane, the denominator planes follow it
        FracPlanes leaves() const {
            if (!leaf_buf.p) throw OvmFail(1, "internal: the leaf layer is not resident");
            return p_base ? FracPlanes{leaf_buf.p, leaf_stride, leaf_stride} : FracPlanes::make(leaf_buf.p, leaf_stride);
        }
        // planes positioned at element 0 of my window of layer j
        FracPlanes layer_planes(unsigned j) const { return j == TR ? leaves() : tree().at(toff[j]); }
        // planes positioned at global fraction i of layer j (i inside my window)
        FracPlanes layer_at(unsigned j, uint64_t i) const { return layer_planes(j).at(i - fb[j]); }
        uint64_t leaf_origin() const { return fb[TR]; }
        FracPlanes recs(unsigned sr) const { return FracPlanes::make(tbuf[sr] ? rec_b.p : rec_a.p, wcap); }

        // my record ranges of a layer's tables: chain[0] = records of the layer, chain[s + 1] = up(chain[s])
        std::vector<URange> chain_for(unsigned round) const {
            std::vector<URange> c(round + 2);
            c[0] = up(plan.F[round + 1][me]);
            for (unsigned s2 = 0; s2 <= round; s2++) c[s2 + 1] = up(c[s2]);
            return c;
        }
        EqHeapPlan heap_plan_for(unsigned round) const {
            const std::vector<URange> c = chain_for(round);
            std::vector<std::pair<uint64_t, uint64_t>> need(round);
            for (unsigned d = 0; d < round; d++) need[d] = {c[round - d].lo, c[round - d].hi};  // pairs of sub-round round-1-d
            return eq_heap_plan(round, need);
        }

        // limb planes 1..3 of my leaf numerators (small trees only: big ones never read them)
        void clear_leaf_numerators() {
            const URange &r = plan.F[TR][me];
            if (p_base || r.empty()) return;
            for (int k = 1; k < 4; k++)
                OVM_THROW_IF(hipMemsetAsync(leaves().p + k * leaf_stride + (r.lo - fb[TR]), 0, r.size() * sizeof(uint32_t), pr.cx.stream));
        }
        void build() {
            for (unsigned j = TR; j-- > 0;) {
                const URange &r = plan.F[j][me];
                if (r.empty()) continue;
                OVM_THROW_IF(gkr_build_layer(layer_at(j + 1, 2 * r.lo), layer_at(j, r.lo), r.size(), p_base && j + 1 == TR, pr.cx.stream));
            }
        }
        void read_layer_edges(std::vector<std::array<Frac, 2>> &out) override {
            // (the leaf layer lives in its own buffer: its two edge fractions are a list of their own)
            for (int part = 0; part < 2; part++) {
                std::vector<uint64_t> pos;
                std::vector<uint32_t> pb;
                std::vector<std::pair<unsigned, int>> where;
                for (unsigned j = part ? TR : 0; j < (part ? TR + 1 : TR); j++) {
                    const URange &r = plan.F[j][me];
                    if (r.empty()) continue;
                    for (int e = 0; e < 2; e++) {
                        pos.push_back((j == TR ? 0 : toff[j]) + ((e ? r.hi - 1 : r.lo) - fb[j]));
                        pb.push_back(p_base && j == TR ? 1 : 0);
                        where.push_back({j, e});
                    }
                }
                if (pos.empty()) continue;
                const FracPlanes planes = part ? leaves() : tree();
                for (size_t at = 0; at < pos.size(); at += 32) {
                    const uint32_t n = (uint32_t)std::min<size_t>(32, pos.size() - at);
                    auto d_pos = pr.cx.upload(std::vector<uint64_t>(pos.begin() + at, pos.begin() + at + n));
                    auto d_pb = pr.cx.upload(std::vector<uint32_t>(pb.begin() + at, pb.begin() + at + n));
                    OVM_THROW_IF(gkr_read_list(planes, d_pos.p, d_pb.p, n, pr.mb.arm(nullptr), pr.cx.stream));
                    auto v = pr.mb.take(2 * n, pr.cx.stream);
                    for (uint32_t i = 0; i < n; i++) out[where[at + i].first][where[at + i].second] = Frac{v[2 * i], v[2 * i + 1]};
                }
            }
            // memory-bounded mode: every layer above exists and the edges are with the host -- the leaves can go
            // (the mailbox reads above have drained the stream: nothing in flight reads them)
            if (drop_leaves) leaf_buf.release();
        }
        void begin_layer(unsigned round, const std::vector<EF> &pt_in) override {
            if (drop_leaves && round == 1) alloc_round_buffers(TR - 2);
            if (drop_leaves && round + 1 == TR) {
                // the last layer's sumcheck reads the leaves and nothing above them: the rest of the tree makes room
                // for the leaves, which are generated a second time (0.7 ms at config D), and for tables of their size
                OVM_THROW_IF(hipStreamSynchronize(pr.cx.stream));
                tree_buf.release();
                alloc_round_buffers(TR - 1);
                alloc_leaves();
                clear_leaf_numerators();
                make_leaves();
            }
            Rchain = chain_for(round);
            hp = heap_plan_for(round);
            point = pt_in;
            mat_sr = 0;
            tbuf.assign(round + 2, 0);
            r_hist.assign(round + 2, EF::zero());
            la.valid = false;
            rbase.assign(round + 1, 0);
            for (unsigned sr = 1; sr <= round; sr++) rbase[sr] = (2 * Rchain[sr + 1].lo) & ~uint64_t(31);
            split_m = 0;
            if (hp.empty) return;
            if (tuning().gkr_split_eq && round >= tuning().gkr_split_min_round) {
                // Two small heaps instead of the layer's whole one (device_common.h: EqSplit): variables [0, M) of the
                // point and variables [M, round). Depth d of a heap = the table over the LAST d entries of its point, so
                // the hi heap is built on point[M-1 ..) (its first entry belongs to no table) and its depth round - M is
                // the full hi table; sub-round sr takes depth M - sr - 1 of the lo heap.
                split_m = round / 2;
                auto full = [](uint32_t n) {
                    std::vector<std::pair<uint64_t, uint64_t>> need(n);
                    for (uint32_t d = 0; d < n; d++) need[d] = {0, uint64_t(1) << d};
                    return need;
                };
                hp_lo = eq_heap_plan(split_m, full(split_m));
                hp_hi = eq_heap_plan(round - split_m + 1, full(round - split_m + 1));
                auto fit = [&](DevBuf<uint32_t> &buf, uint64_t &stride, const EqHeapPlan &pl) {
                    const uint64_t need = up64(pl.total) + PLANE_SKEW;
                    if (stride < need) {
                        stride = need;
                        buf.alloc(4 * stride);
                    }
                };
                fit(eq_lo_buf, eq_lo_stride, hp_lo);
                fit(eq_hi_buf, eq_hi_stride, hp_hi);
                OVM_THROW_IF(eq_heap_run(point.data(), hp_lo, EFPlanes{eq_lo_buf.p, eq_lo_stride}, pr.cx.stream));
                OVM_THROW_IF(eq_heap_run(point.data() + split_m - 1, hp_hi, EFPlanes{eq_hi_buf.p, eq_hi_stride}, pr.cx.stream));
                return;
            }
            OVM_THROW_IF(eq_heap_run(point.data(), hp, EFPlanes{eq_buf.p, eq_stride}, pr.cx.stream));
        }
        // planes positioned at record y of table(sr) (sr == 0: the tree layer)
        FracPlanes table_at(unsigned round, unsigned sr, uint64_t y) const {
            if (sr == 0) return layer_at(round + 1, 2 * y);
            return recs(sr).at(2 * (y - rbase[sr]));
        }
        // the weights of sub-round sr's pairs, positioned at global pair y_first
        EqSplit eq_for(unsigned round, unsigned sr, uint64_t y_first) const {
            const unsigned d = round - 1 - sr;
            EqSplit eq{EFPlanes{nullptr, 0}, EFPlanes{nullptr, 0}, EQ_WHOLE, y_first};
            if (!split_m) {
                eq.lo = EFPlanes{eq_buf.p + hp.at(d, y_first), eq_stride};
                return eq;
            }
            const unsigned n_hi = round - split_m;
            if (sr + 1 < split_m) {
                const unsigned d_lo = split_m - sr - 1;
                eq.lo = EFPlanes{eq_lo_buf.p + hp_lo.at(d_lo, 0), eq_lo_stride};
                eq.hi = EFPlanes{eq_hi_buf.p + hp_hi.at(n_hi, 0), eq_hi_stride};
                eq.lo_bits = d_lo;
            } else {
                eq.hi = EFPlanes{eq_hi_buf.p + hp_hi.at(d, 0), eq_hi_stride};
                eq.lo_bits = 0;
            }
            return eq;
        }
        void subround(unsigned round, unsigned sr, URange pairs, const uint64_t *edge_idx, unsigned n_edges, EF r_prev, EF lambda,
                      unsigned n_points, EF *S, GkrRec *edges) override {
            hipStream_t st = pr.cx.stream;
            const bool have_pairs = !pairs.empty();
            if (sr > 0) r_hist[sr - 1] = r_prev;
            // ---- served from the lookahead of the pass before: no launch, no round trip ----
            if (la.valid && la.sr == sr && have_pairs && pairs.lo == la.pairs.lo && pairs.hi == la.pairs.hi && n_points == 2 && n_edges <= 2) {
                bool ok = true;
                for (unsigned k = 0; k < n_edges && ok; k++) {
                    ok = false;
                    for (unsigned j = 0; j < la.n_src; j++)
                        if (la.src_pair[j] == edge_idx[k]) {
                            edges[k] = rec_fold(la.src[j][0], la.src[j][1], r_prev);
                            ok = true;
                        }
                }
                if (ok) {
                    // the quadratic through (0, v0), (1, v1), (2, v2) at r_prev
                    const EF inv2 = EF::from_base(fp(2).inv());
                    const EF rm1 = r_prev - EF::one(), rm2 = rm1 - EF::one();
                    const EF l0 = rm1 * rm2 * inv2, l1 = EF::zero() - r_prev * rm2, l2 = r_prev * rm1 * inv2;
                    for (int x = 0; x < 2; x++) S[x] = la.v[0][x] * l0 + la.v[1][x] * l1 + la.v[2][x] * l2;
                    la.valid = false;
                    la_hits++;
                    return;
                }
            }
            la.valid = false;
            // ---- a pass over the latest table that exists: table(sr) is its 0- / 1- / 2-fold ----
            const unsigned folds = sr - mat_sr;
            if (folds > 2 || (sr == 0) != (folds == 0)) throw OvmFail(1, "internal: GKR sub-round too far from the last written table");
            const EF r1 = folds == 2 ? r_hist[sr - 2] : r_prev, r2 = r_prev;
            const bool src_leaf = p_base && round + 1 == TR && mat_sr == 0;  // the source is the leaf layer (base-field numerators)
            const unsigned L = round + 1;
            if (n_edges) {
                // (first: when a sums kernel follows, it is the one that raises the flag)
                RedOut ro = pr.mb.arm(nullptr, 8);
                if (have_pairs) ro.flag = nullptr;
                // record e of table(sr) = the `folds`-fold of records e 2^folds .. of the source table; the indices are at
                // most two runs of consecutive records
                uint32_t n0 = 1;
                while (n0 < n_edges && edge_idx[n0] == edge_idx[n0 - 1] + 1) n0++;
                const uint32_t n1 = n_edges - n0;
                for (uint32_t k = n0 + 1; k < n_edges; k++)
                    if (edge_idx[k] != edge_idx[k - 1] + 1) throw OvmFail(1, "internal: more than two runs of edge records");
                const FracPlanes base = mat_sr == 0 ? layer_planes(L) : recs(mat_sr);
                auto at_of = [&](uint64_t e) -> uint64_t {
                    const uint64_t src_rec = e << folds;  // first source record
                    return mat_sr == 0 ? 2 * src_rec - fb[L] : 2 * (src_rec - rbase[mat_sr]);
                };
                OVM_THROW_IF(gkr_edge_records(base, at_of(edge_idx[0]), n1 ? at_of(edge_idx[n0]) : 0, n0, n1, src_leaf, (int)folds, r1, r2, ro, st));
            }
            bool use_la = false;
            if (have_pairs) {
                const uint64_t y0 = pairs.lo, n_pairs = pairs.size();
                // lookahead: sub-round sr + 1 must be an ordinary one (it exists, my table is still above the host-tail size
                // there, two evaluation points) and ask for exactly the pairs the chain predicts
                use_la = tuning().gkr_lookahead && !tuning().gkr_three_points && n_points == 2 && sr + 1 < round &&
                         sr + 2 < Rchain.size() && Rchain[sr + 1].lo == pairs.lo && Rchain[sr + 1].hi == pairs.hi &&
                         Rchain[sr + 1].size() > GKR_HOST_TAIL && !Rchain[sr + 2].empty() &&
                         n_pairs <= (uint64_t(1) << std::min(tuning().gkr_la_max_log, 40u));
                uint64_t want[2] = {~0ull, ~0ull};
                EqSplit eq_n = eq_for(round, sr, y0);
                if (use_la) {
                    la.n_src = range_edges(Rchain[sr + 1], la.src_pair);  // sub-round sr + 1's unpaired records = pairs of table(sr)
                    for (unsigned k = 0; k < la.n_src; k++) want[k] = la.src_pair[k] - y0;
                    eq_n = eq_for(round, sr + 1, Rchain[sr + 2].lo);
                    if (Rchain[sr + 2].lo != (y0 + 1) / 2) throw OvmFail(1, "internal: GKR lookahead range mismatch");
                }
                const FracPlanes in = mat_sr == 0 ? layer_at(L, (2 * y0) << (folds + 1)) : recs(mat_sr).at(2 * (((2 * y0) << folds) - rbase[mat_sr]));
                FracPlanes out = in;
                if (folds) {
                    tbuf[sr] = mat_sr == 0 ? 0 : (uint8_t)(1 - tbuf[mat_sr]);
                    out = recs(sr).at(2 * (2 * y0 - rbase[sr]));
                }
                // (the plain one-fold passes keep the kernels of rounds 1-3: the generic pass kernel measured 0.34 ms per proof
                // slower in that role -- 4.90 against 4.56 ms over the 189 fold passes of config D, profiles/r04/)
                if (!use_la && folds == 0)
                    OVM_THROW_IF(gkr_round(in, src_leaf, eq_for(round, sr, y0), n_pairs, lambda, n_points, pr.mb.arm(partials.p), st));
                else if (!use_la && folds == 1)
                    OVM_THROW_IF(gkr_fold_round(in, src_leaf, eq_for(round, sr, y0), out, n_pairs, r1, lambda, n_points, pr.mb.arm(partials.p), st));
                else
                    OVM_THROW_IF(gkr_pass(in, src_leaf, (int)folds, eq_for(round, sr, y0), eq_n, out, n_pairs, r1, r2, lambda, n_points, use_la,
                                          use_la ? point[sr + 1] : EF::zero(), want[0], want[1], pr.mb.slot_ptr(LA_DUMP_SLOT),
                                          pr.mb.arm(partials.p), st));
                (use_la ? la_passes : plain_passes)++;
            }
            pr.mb.wait(st);
            if (have_pairs) {
                auto v = pr.mb.read(0, use_la ? 8 : n_points);
                for (unsigned k = 0; k < n_points; k++) S[k] = v[k];
                if (folds) mat_sr = sr;
                if (use_la) {
                    la.valid = true;
                    la.sr = sr + 1;
                    la.pairs = Rchain[sr + 2];
                    for (int j = 0; j < 3; j++)
                        for (int x = 0; x < 2; x++) la.v[j][x] = v[2 + 2 * j + x];
                    if (la.n_src) {
                        auto d = pr.mb.read(LA_DUMP_SLOT, 16);
                        for (unsigned k = 0; k < la.n_src; k++)
                            for (int h2 = 0; h2 < 2; h2++)
                                for (int c = 0; c < 4; c++) la.src[k][h2].v[c] = d[8 * k + 4 * h2 + c];
                    }
                }
            }
            if (n_edges) {
                auto v = pr.mb.read(8, 4 * n_edges);
                for (unsigned k = 0; k < n_edges; k++)
                    for (int c = 0; c < 4; c++) edges[k].v[c] = v[4 * k + c];
            }
        }
    };

    // -----------------------------------------------------------------------------------------
    // Stage 2: zerocheck + LogUp (MultiRapProver::prove_rap_constraints)
    // -----------------------------------------------------------------------------------------
    // Per trace: heights, the column map of its view (selectors, preprocessed, cached mains, common main, each with
    // its rotated twin when the AIR needs it) -- on every rank -- and, on the owner, the device descriptors.
    // imeta / total_interactions: the interaction layout's input (one entry per trace).
    std::vector<TraceView> build_trace_views(uint64_t &total_interactions, std::vector<std::pair<size_t, unsigned>> &imeta) {
        const unsigned l_skip = P().l_skip;
        const size_t T = traces.size();
        std::vector<TraceView> tv(T);
        for (size_t t = 0; t < T; t++) {
            TraceView &v = tv[t];
            v.log_height = log2_strict(traces[t].height);
            v.log_lifted = std::max(v.log_height, l_skip);
            v.n = (int)v.log_height - (int)l_skip;
            v.n_lift = (unsigned)std::max(v.n, 0);
            v.cm.need_rot = air(t).need_rot;
            v.cm.pre_width = air(t).preprocessed_width;
            v.cm.part_width = air(t).cached_widths;
            v.cm.part_width.push_back(air(t).common_main_width);
            v.cm.finish();
            size_t ni = air(t).interactions.size();
            total_interactions += (uint64_t)ni << v.log_lifted;
            imeta.push_back({ni, v.log_lifted});
            if (!mine(t)) continue;
            // device view: selectors + columns
            v.sels.alloc(size_t(3) << v.log_lifted);
            OVM_THROW_IF(build_selectors(v.sels.p, v.log_lifted, v.log_height, cx.stream));
            std::vector<ColDesc> cd(v.cm.total);
            std::vector<int32_t> twin(v.cm.total, -1);
            for (int sidx = 0; sidx < 3; sidx++) cd[sidx] = ColDesc{v.sels.p + ((size_t)sidx << v.log_lifted), v.log_lifted, 0};
            const size_t h = traces[t].height;
            auto put_mat = [&](const Fp *ptr, size_t w, uint32_t base) {
                for (size_t c = 0; c < w; c++) {
                    cd[base + c] = ColDesc{ptr + c * h, v.log_height, 0};
                    if (v.cm.need_rot) {
                        cd[base + w + c] = ColDesc{ptr + c * h, v.log_height, 1};
                        // (a chunk and its rotated view share cells only when the trace has whole 16-row chunks)
                        if (v.log_height >= 4) twin[base + c] = (int32_t)(base + w + c), twin[base + w + c] = -2;
                    }
                }
            };
            if (v.cm.pre_width) {
                const PcsData *pp = pre_of(t);
                if (!pp || pp->mat_h != h || pp->mat_w != v.cm.pre_width)
                    throw OvmFail(1, "preprocessed trace missing or of the wrong shape");
                put_mat(pp->mat_ptr, pp->mat_w, v.cm.pre_col(0, 0));
            }
            const size_t n_cached = air(t).cached_widths.size();
            if (traces[t].cached.size() != n_cached) throw OvmFail(1, "cached main partitions: wrong count");
            for (size_t k = 0; k < n_cached; k++) {
                const PcsData *cp = traces[t].cached[k];
                if (!cp || cp->mat_h != h || cp->mat_w != air(t).cached_widths[k] || !cp->mat_ptr)
                    throw OvmFail(1, "cached main partition of the wrong shape");
                put_mat(cp->mat_ptr, cp->mat_w, v.cm.col((uint32_t)k, 0, 0));
            }
            put_mat(traces[t].d_trace, air(t).common_main_width, v.cm.col((uint32_t)n_cached, 0, 0));
            v.cols = cx.upload(cd);
            v.twins = cx.upload(twin);
        }
        return tv;
    }

    // LogUp-GKR (prover/logup_zerocheck/fractional_sumcheck_gkr.rs:60-213): this rank's leaves on its device, the
    // tree over them, the layer sumchecks through the driver of gkr_shard.h (padding and cross-rank records on the host).
    GkrOutput prove_gkr(const std::vector<TraceView> &tv, std::vector<TracePrograms> &progs, const Layout &ilayout, unsigned TR,
                        EF alpha, const std::vector<EF> &beta_pows, const std::function<void()> &after_build) {
        const size_t T = traces.size();
        const uint64_t n_leaves = uint64_t(1) << TR;
        // The interaction slices are packed from row 0 with no gaps (descending powers of two) in sorted
        // trace order, so the leaves of a rank (a block of consecutive traces) are one run; what is left
        // above the last slice is padding.
        uint64_t used_leaves = 0;
        std::vector<URange> rank_leaves(sh.world);
        for (size_t t = 0; t < T; t++)
            for (size_t k = 0; k < air(t).interactions.size(); k++) {
                const Slice &sl = ilayout.cols[ilayout.mat_starts[t] + k].slice;
                const uint64_t lo = sl.row_idx, hi = sl.row_idx + (uint64_t(1) << sl.log_height);
                used_leaves = std::max(used_leaves, hi);
                URange &r = rank_leaves[owner[t]];
                r = r.empty() ? URange{lo, hi} : URange{std::min(r.lo, lo), std::max(r.hi, hi)};
            }
        if (used_leaves > n_leaves) throw OvmFail(1, "internal: interaction layout exceeds the tree");
        const GkrPlan plan(TR, rank_leaves, used_leaves);
        // big trees: only limb 0 of the leaf numerators exists (small ones are read through generic paths)
        const bool p_base = TR >= 12;
        GkrDevice gd(*this, plan, p_base);
        // What the leaf kernels of this rank's traces need, kept until the tree no longer wants its leaves (the
        // memory-bounded mode launches them a second time); traces of fewer than 2^20 rows share one launch.
        struct LeafKeep {
            ProgDev pd;
            DevBuf<EF> lc;
            DevBuf<uint64_t> lb;
            DevBuf<uint32_t> ll;
            Fp norm;
            size_t t;
            bool batched;
        };
        auto leaf_keep = std::make_shared<std::vector<LeafKeep>>();
        auto leaf_jobs = std::make_shared<std::vector<LeavesJob>>();
        for (size_t t = 0; t < T; t++) {
            const AirKey &a = air(t);
            if (a.interactions.empty() || !mine(t)) continue;
            std::vector<EF> leaf_const;
            std::vector<uint64_t> leaf_base;
            std::vector<uint32_t> leaf_log_len;
            for (size_t k = 0; k < a.interactions.size(); k++) {
                const Interaction &it = a.interactions[k];
                leaf_const.push_back(beta_pows[it.message.size()] * fp((uint64_t)it.bus_index + 1) + alpha);
                const Slice &s = ilayout.cols[ilayout.mat_starts[t] + k].slice;
                leaf_base.push_back(s.row_idx - gd.leaf_origin());  // (relative to this rank's leaf window)
                leaf_log_len.push_back(s.log_height);
            }
            auto d_lc = cx.upload(leaf_const);
            auto d_lb = cx.upload(leaf_base);
            auto d_ll = cx.upload(leaf_log_len);
            Fp norm = fp(uint64_t(1) << (tv[t].log_lifted - tv[t].log_height)).inv();
            // staged for one row per lane (log_lifted 0 selects that plan)
            ProgDev pd = make_prog(std::move(progs[t].leaves), beta_pows, true, 0, 1);
            const bool batched = pd.staged.blocks && tv[t].log_height < 20;
            if (batched) {
                LeavesJob j{};
                j.prog = pd.staged, j.leaf_const = d_lc.p, j.cols = tv[t].cols.p, j.leaf_base = d_lb.p;
                j.leaf_log_len = d_ll.p, j.norm = norm, j.log_height = tv[t].log_height;
                j.gx = (uint32_t)ceil_div(1ull << tv[t].log_height, 256);
                j.wg0 = leaf_jobs->empty() ? 0 : leaf_jobs->back().wg0 + leaf_jobs->back().gx;
                leaf_jobs->push_back(j);
            }
            leaf_keep->push_back(LeafKeep{std::move(pd), std::move(d_lc), std::move(d_lb), std::move(d_ll), norm, t, batched});
        }
        gd.make_leaves = [this, &gd, &tv, leaf_keep, leaf_jobs]() {
            for (const LeafKeep &k : *leaf_keep)
                if (!k.batched)
                    OVM_THROW_IF(gkr_eval_leaves(k.pd.dev(), k.pd.staged, k.lc.p, tv[k.t].cols.p, tv[k.t].log_height, k.lb.p, k.ll.p,
                                                 k.norm, gd.leaves(), cx.stream));
            if (!leaf_jobs->empty()) {
                auto d_jobs = cx.upload(*leaf_jobs);
                OVM_THROW_IF(gkr_eval_leaves_batched(d_jobs.p, leaf_jobs->data(), (uint32_t)leaf_jobs->size(), gd.leaves(), cx.stream));
            }
        };
        {
            OVM_PROF("gkr.leaves", cx.stream);
            gd.clear_leaf_numerators();
            gd.make_leaves();
            // no synchronisation: the temporaries go back to the block cache and are only ever reused
            // by later work on this same stream
            if (!gd.drop_leaves) leaf_keep->clear(), leaf_jobs->clear();
        }
        {
            OVM_PROF("gkr.build_tree", cx.stream);
            wt.mark("rap.gkr.leaves");
            gd.build();
        }
        wt.mark("rap.gkr.build");
        if (after_build) after_build();  // (the tree is enqueued: what follows is latency-bound for its first dozen layers)
        OVM_PROF("gkr.layer_sumchecks", cx.stream);
        GkrSharded drv(plan, sh, gd, alpha);
        drv.force_three_points = force_three_points;
        drv.observe = [this](const EF &e) { ts.observe_ext(e); };
        drv.sample = [this]() { return ts.sample_ext(); };
        GkrOutput gout = drv.run();
        if (tuning().trace)
            fprintf(stderr, "[ovm] gkr sub-rounds on this rank: %llu passes with lookahead, %llu plain passes, %llu served from a lookahead\n",
                    (unsigned long long)gd.la_passes, (unsigned long long)gd.plain_passes, (unsigned long long)gd.la_hits);
        return gout;
    }

    // The univariate round of the batch sumcheck, host algebra on a few dozen extension-field elements
    // (reference: cpu-backend/src/logup_zerocheck/mod.rs:1468-1563): sp0[2t], sp0[2t+1] = the LogUp numerator /
    // denominator polynomials of trace t, sp0[2T + t] its zerocheck quotient part. Observes the per-trace sum
    // claims, samples mu, observes the coefficients of s_0.
    struct S0 {
        std::vector<EF> numer_terms, denom_terms, mu_pows, s0;
    };
    S0 assemble_s0(const std::vector<std::vector<EF>> &sp0, const std::vector<EF> &xi) {
        const unsigned l_skip = P().l_skip, cdeg_global = P().max_constraint_degree;
        const size_t N = size_t(1) << l_skip, T = traces.size();
        const unsigned sp0_deg = cdeg_global * (unsigned)(N - 1), s_deg = cdeg_global + 1, s0_deg = s_deg * (unsigned)(N - 1);
        const size_t large = size_t(1) << log2_ceil(s0_deg + 1);
        // product with the degree-(N-1) eq polynomial. The reference multiplies on a domain of size
        // `large` (no wrap-around: deg <= s0_deg < large), i.e. a plain polynomial product; with tens of
        // AIRs the N-term schoolbook product is much cheaper on the host than two size-`large` DFTs.
        auto mul_by = [&](const std::vector<EF> &a, const std::vector<EF> &eq_coeffs) {
            std::vector<EF> out(large, EF::zero());
            for (size_t j = 0; j < eq_coeffs.size(); j++) {
                if (is_zero(eq_coeffs[j])) continue;
                const EF::Pre e = eq_coeffs[j].pre();
                for (size_t i = 0; i < a.size() && i + j < large; i++) out[i + j] += a[i] * e;
            }
            return out;
        };
        std::vector<std::vector<EF>> s0_logup(2 * T);
        {
            std::vector<EF> eqc = small_dft(eq_hypercube(std::vector<EF>(xi.begin(), xi.begin() + l_skip)), true);
            for (size_t i = 0; i < 2 * T; i++) {
                std::vector<EF> c(sp0[i].begin(), sp0[i].begin() + std::min(sp0[i].size(), (size_t)sp0_deg + 1));
                s0_logup[i] = mul_by(c, eqc);
            }
        }
        std::vector<EF> numer_terms, denom_terms;
        for (size_t t = 0; t < T; t++) {
            EF sums[2];
            for (int dnm = 0; dnm < 2; dnm++) {
                EF s = EF::zero();
                for (size_t j = 0; j <= s0_deg; j += N) s += s0_logup[2 * t + dnm][j];
                sums[dnm] = s * fp(N);
            }
            ts.observe_ext(sums[0]);
            ts.observe_ext(sums[1]);
            numer_terms.push_back(sums[0]);
            denom_terms.push_back(sums[1]);
        }
        EF mu = ts.sample_ext();
        std::vector<EF> mu_pows = powers_of(mu, 3 * T);
        std::vector<EF> s0_zc;
        {
            Fp n_inv = Fp::one().halve().pow(l_skip);
            std::vector<EF> eqc(N);
            EF xp = xi[0];
            for (size_t i = 0; i < N; i++) {
                eqc[i] = xp * n_inv;
                xp = xp * xi[0];
            }
            std::reverse(eqc.begin(), eqc.end());
            eqc[0] = ef(n_inv);
            std::vector<EF> c(sp0_deg + 1, EF::zero());
            for (size_t j = 0; j <= sp0_deg; j++)
                for (size_t t = 0; t < T; t++)
                    if (j < sp0[2 * T + t].size()) c[j] += mu_pows[2 * T + t] * sp0[2 * T + t][j];
            s0_zc = mul_by(c, eqc);
        }
        std::vector<EF> s0(s0_deg + 1);
        for (size_t j = 0; j <= s0_deg; j++) {
            EF c = s0_zc[j];
            for (size_t i = 0; i < 2 * T; i++) c += mu_pows[i] * s0_logup[i][j];
            ts.observe_ext(c);
            s0[j] = c;
        }
        return S0{std::move(numer_terms), std::move(denom_terms), std::move(mu_pows), std::move(s0)};
    }

    std::vector<EF> prove_rap_constraints() {
        std::vector<TracePrograms> progs = programs_future.valid() ? programs_future.get() : compile_all_programs();
        const unsigned l_skip = P().l_skip;
        const size_t N = size_t(1) << l_skip, T = traces.size();
        const unsigned cdeg_global = P().max_constraint_degree;
        uint64_t total_interactions = 0;
        std::vector<std::pair<size_t, unsigned>> imeta;
        std::vector<TraceView> tv = build_trace_views(total_interactions, imeta);
        const unsigned n_max = tv[0].n_lift;
        unsigned n_logup = 0;
        if (total_interactions) n_logup = (64 - (unsigned)__builtin_clzll(total_interactions)) - l_skip;
        Layout ilayout(0, l_skip + n_logup, imeta);

        ByteWriter &w = out;
        Fp logup_pow = ts.grind(P().logup_pow_bits);
        EF alpha = ts.sample_ext(), beta = ts.sample_ext();
        size_t max_msg = 0, max_constraints = 0;
        for (size_t t = 0; t < T; t++)
            for (auto &it : air(t).interactions) max_msg = std::max(max_msg, it.message.size());
        for (auto &a : pk.airs) max_constraints = std::max(max_constraints, a.constraint_idx.size());
        std::vector<EF> beta_pows = powers_of(beta, max_msg + 1);

        // shift^i / N per coset (the iDFT scale is folded in): the same for every trace
        const Fp g = fp(bb::GENERATOR), omega_skip = bb::two_adic_generator(l_skip);
        std::vector<Fp> shift_pows(MAX_COSETS * N);
        for (unsigned k = 0; k < MAX_COSETS; k++) {
            Fp sh = g.pow(k + 1), cur = fp((uint64_t)N).inv();
            for (size_t i = 0; i < N; i++) {
                shift_pows[k * N + i] = cur;
                cur *= sh;
            }
        }
        auto d_sp_all = cx.upload(shift_pows);

        // ---- round 0's coset LDE, early ----
        // The extension of a trace to the cosets depends on no challenge. Once the GKR tree is enqueued the device
        // has ~3 ms of layer sumchecks ahead in which it mostly waits for the host (tables of a few pairs, one mailbox
        // round trip per sub-round): the LDE (1.6 ms at config D) runs beside them on the library's low-priority side
        // stream and round 0 finds it done. Not with zerocheck_save_memory (the LDE buffer is a slab there, reused),
        // and not when the buffer cannot be had (round 0 then allocates it after the tree has been freed).
        struct EarlyLde {
            DevBuf<Fp> lde;
            DevBuf<uint32_t> d_used;
            std::vector<uint32_t> used;
            unsigned n_cosets = 0;
            bool done = false;
        };
        struct EarlyLdeSet {  // (a proof that fails between fork and join must not hand the buffers back while the kernel runs)
            std::vector<EarlyLde> per_trace;
            bool in_flight = false;
            ~EarlyLdeSet() {
                if (in_flight) (void)hipStreamSynchronize(side_stream().s);
            }
        } early;
        early.per_trace.resize(T);
        auto used_columns = [&](size_t t, const std::vector<uint32_t> &zc_used, const std::vector<uint32_t> *lg_used) {
            std::vector<char> usedm(tv[t].cm.total, 0);
            for (auto c : zc_used) usedm[c] = 1;
            if (lg_used)
                for (auto c : *lg_used) usedm[c] = 1;
            std::vector<uint32_t> used;
            for (size_t c = 0; c < usedm.size(); c++)
                if (usedm[c]) used.push_back((uint32_t)c);
            return used;
        };
        std::function<void()> enqueue_early_lde;
        if (tuning().early_lde && !prover_config().zerocheck_save_memory) {
            for (size_t t = 0; t < T; t++) {
                if (!mine(t)) continue;
                const AirKey &a = air(t);
                EarlyLde &e = early.per_trace[t];
                const unsigned cdeg = a.max_constraint_degree;
                e.n_cosets = std::max(cdeg ? cdeg - 1 : 0u, a.interactions.empty() ? 0u : cdeg);
                if (!e.n_cosets || e.n_cosets > MAX_COSETS) continue;
                e.used = used_columns(t, progs[t].zc.used_cols, a.interactions.empty() ? nullptr : &progs[t].lg.used_cols);
            }
            enqueue_early_lde = [&]() {
                SideStream &ss = side_stream();
                try {
                    ss.init();
                } catch (const OvmFail &) {
                    return;  // (no second stream to be had: round 0 extends the traces where it uses them)
                }
                bool any = false;
                for (size_t t = 0; t < T; t++) {
                    EarlyLde &e = early.per_trace[t];
                    if (e.used.empty()) continue;
                    try {
                        e.lde.alloc((size_t)e.n_cosets * tv[t].cm.total << tv[t].log_lifted);
                    } catch (const OvmFail &) {
                        continue;
                    }
                    e.d_used = cx.upload(e.used);
                    e.done = true;
                    any = true;
                }
                if (!any) return;
                flush_uploads(cx.stream);
                OVM_THROW_IF(hipEventRecord(ss.fork, cx.stream));
                OVM_THROW_IF(hipStreamWaitEvent(ss.s, ss.fork, 0));
                early.in_flight = true;
                {
                    OVM_PROF("round0.coset_lde", ss.s);
                    for (size_t t = 0; t < T; t++) {
                        EarlyLde &e = early.per_trace[t];
                        if (!e.done) continue;
                        OVM_THROW_IF(coset_lde(tv[t].cols.p, e.d_used.p, (uint32_t)e.used.size(), (uint32_t)tv[t].cm.total, tv[t].log_lifted,
                                               l_skip, e.n_cosets, d_sp_all.p, e.lde.p, ss.s, 0, tuning().early_lde_lds_kib << 10));
                    }
                }
                OVM_THROW_IF(hipEventRecord(ss.join, ss.s));
            };
        }

        // ---- GKR ----
        const unsigned TR = l_skip + n_logup;
        GkrOutput gk;
        if (!ilayout.cols.empty()) gk = prove_gkr(tv, progs, ilayout, TR, alpha, beta_pows, enqueue_early_lde);
        std::vector<EF> xi = std::move(gk.xi);
        const EF q0_claim = gk.q0_claim;
        const std::vector<std::array<EF, 4>> &claims = gk.claims;               // p0,p1,q0,q1 per layer
        const std::vector<std::vector<std::array<EF, 3>>> &gkr_polys = gk.polys;
        wt.mark("rap.gkr");
        const unsigned n_global = std::max(n_max, n_logup);
        while (xi.size() != l_skip + n_global) xi.push_back(ts.sample_ext());
        // l_skip = 0 with every trace one row tall and no interactions: xi is EMPTY and the protocol has no round-0 variable
        // to bind -- the reference indexes `prover.xi[0]` and panics (prover/logup_zerocheck/mod.rs:255); refused here.
        if (xi.empty()) throw OvmFail((int)hipErrorInvalidValue, "degenerate context: l_skip = 0 and no trace taller than one row (xi is empty; the reference panics at logup_zerocheck/mod.rs:255)");

        // gkr proof bytes
        w.f(logup_pow);
        w.e(q0_claim);
        w.usize(claims.size());
        for (auto &c : claims)
            for (auto &x : c) w.e(x);
        for (auto &round : gkr_polys)
            for (auto &arr : round)
                for (auto &x : arr) w.e(x);
        section_offsets.push_back(w.buf.size());

        EF lambda = ts.sample_ext();
        std::vector<EF> lambda_pows = powers_of(lambda, std::max<size_t>(max_constraints, 1));
        std::vector<Fp> omega_pows = powers_of(omega_skip, N);

        // ---- programs, eq weights, round 0 ----
        struct GroupedDev {
            GroupedProgram gp;
            DevBuf<Instr> code;
            DevBuf<uint32_t> off;
        };
        auto make_grouped = [&](GroupedProgram gp) {
            auto g = std::make_unique<GroupedDev>();
            g->code = cx.upload(gp.code);
            g->off = cx.upload(gp.off);
            g->gp = std::move(gp);
            return g;
        };
        struct TraceProgs {
            std::unique_ptr<ProgDev> zc, lg;
            std::unique_ptr<GroupedDev> zc_g, lg_g;
            std::unique_ptr<StagedGroups> zc_sg, lg_sg;  // round 0 of long programs on short traces
            std::vector<EF> zc_w, lg_w;
            EF lg_const = EF::zero();
            std::vector<EF> eq3b;
        };
        std::vector<TraceProgs> tp(T);
        std::vector<std::vector<EF>> sp0(3 * T);
        // Phase A enqueues every trace's work (eq table, coset LDE, zerocheck and logup evaluation)
        // without waiting; each reduction gets its own partial-sum buffer and mailbox slots. Phase B
        // waits ONCE and does the host algebra. With tens of AIRs per proof a wait per AIR per
        // reduction was most of this stage's time.
        struct Round0Job {
            unsigned cdeg = 0, zc_cosets = 0, lg_cosets = 0;
            size_t zc_slot = 0, lg_slot = 0, zc_stride = 0, lg_stride = 0;
            uint32_t n_slabs = 1;
            DevBuf<EF> d_eq, zc_partials, lg_partials;
            DevBuf<Fp> lde;
            DevBuf<uint32_t> d_used;
            DevBuf<Fp> d_sp;
        };
        std::vector<Round0Job> jobs(T);
        // Staged round-0 evaluations of traces of fewer than 2^R0_BATCH_MAX_LOG rows are collected per
        // (rows per lane, outputs) and run in one launch each after the loop: a many-AIR proof has dozens of
        // them, each a few workgroups (OVM_R0_BATCH=0: every trace its own launch).
        static constexpr uint32_t R0_BATCH_MAX_LOG = 20;
        // (as many groups as keep the job within this many workgroups: on taller traces the lanes are there
        // already, and the job's last workgroup sums workgroups x groups records)
        static constexpr uint32_t R0_GROUP_JOB_WGS = 512;
        const bool r0_batch = tuning().r0_batch;
        std::vector<R0Job> r0_groups[3][2];  // [log2 rows][n_out - 1]
        // long programs (more groups than the OVM_R0_GROUPS_MIN threshold, default: more than MLE_GROUPS, i.e.
        // > ~2 K instructions) run their groups as extra workgroups of the job
        const uint32_t r0_groups_min = tuning().r0_groups_min >= 0 ? (uint32_t)tuning().r0_groups_min : MLE_GROUPS + 1;
        auto r0_batchable = [&](const DevStaged &staged, uint32_t log_lifted) {
            // (a skip domain larger than a workgroup goes through round0_eval, which sizes its grid in whole domains)
            return r0_batch && staged.blocks && log_lifted < R0_BATCH_MAX_LOG && N <= RED_THREADS;
        };
        // workgroups (= partial records) the job will use
        auto r0_job_grid = [&](const DevStaged &staged, const StagedGroups *sg, uint32_t log_lifted) -> uint32_t {
            if (!sg) return round0_staged_grid(staged, log_lifted);
            DevStaged widest = sg->host[0];
            widest.num_rows = sg->max_num_rows;
            return round0_staged_grid(widest, log_lifted) * (uint32_t)sg->host.size();
        };
        auto r0_enqueue = [&](const DevStaged &staged, const StagedGroups *sg, const Fp *lde, uint32_t n_cols_total,
                              uint32_t log_lifted, uint32_t n_cosets, uint32_t n_out, const EF *eq_xi, EF add_const,
                              EF *partials, size_t slot) {
            if (!r0_batchable(staged, log_lifted)) return false;
            R0Job j{};
            j.prog = sg ? sg->host[0] : staged;
            j.lde = lde, j.eq_xi = eq_xi, j.partials = partials, j.out = mb.slot_ptr(slot);
            j.add_const = add_const, j.n_cols_total = n_cols_total, j.log_lifted = log_lifted, j.n_cosets = n_cosets;
            j.gx = r0_job_grid(staged, sg, log_lifted);
            j.n_groups = sg ? (uint32_t)sg->host.size() : 1;
            j.gx_group = j.gx / j.n_groups;
            j.lds_rows = sg ? sg->max_num_rows : staged.num_rows;
            j.progs = sg ? sg->descs.p : nullptr;
            r0_groups[j.prog.rows == 4 ? 2 : j.prog.rows == 2 ? 1 : 0][n_out - 1].push_back(j);
            return true;
        };
        size_t next_slot = 0;
        auto d_xi = cx.upload(xi);
        if (early.in_flight) {  // the early LDEs: round 0's kernels on the caller's stream come after them
            OVM_THROW_IF(hipStreamWaitEvent(cx.stream, side_stream().join, 0));
            early.in_flight = false;
        }
        // zerocheck_save_memory: the slab of every trace. Each slab's sums get mailbox slots of their own (all traces
        // are enqueued before the one wait), so the configured slab size is a LOWER bound: while the slabs of all
        // traces together would not fit the mailbox, the trace with the most slabs gets slabs twice as large.
        std::vector<unsigned> slab_logs(T, 0);
        {
            const ProverConfig &pcfg = prover_config();
            std::vector<size_t> per_slab(T, 0);
            auto round64 = [](size_t x) { return (x + 63) & ~size_t(63); };
            for (size_t t = 0; t < T; t++) {
                slab_logs[t] = tv[t].log_lifted;
                if (!mine(t)) continue;
                const unsigned cdeg = air(t).max_constraint_degree;
                const unsigned zc = cdeg ? cdeg - 1 : 0, lg = air(t).interactions.empty() ? 0 : cdeg;
                per_slab[t] = (zc ? round64(zc * N) : 0) + (lg ? round64(lg * 2 * N) : 0);
                if (pcfg.zerocheck_save_memory) slab_logs[t] = std::min(tv[t].log_lifted, std::max(pcfg.round0_slab_log_rows, l_skip));
            }
            const size_t budget = Mailbox::SLOTS - (4 * MAX_COSETS * N + 128);
            for (;;) {
                size_t total = 0, worst = T;
                for (size_t t = 0; t < T; t++) {
                    total += per_slab[t] << (tv[t].log_lifted - slab_logs[t]);
                    if (per_slab[t] && slab_logs[t] < tv[t].log_lifted &&
                        (worst == T || tv[t].log_lifted - slab_logs[t] > tv[worst].log_lifted - slab_logs[worst]))
                        worst = t;
                }
                if (total <= budget || worst == T) break;
                slab_logs[worst]++;
            }
        }
        for (size_t t = 0; t < T; t++) {
            if (!mine(t)) continue;
            const AirKey &a = air(t);
            TraceView &v = tv[t];
            Round0Job &job = jobs[t];
            unsigned cdeg = a.max_constraint_degree;
            job.cdeg = cdeg;
            // eq(xi_3, b) per interaction
            for (size_t k = 0; k < a.interactions.size(); k++) {
                const Slice &s = ilayout.cols[ilayout.mat_starts[t] + k].slice;
                size_t b_int = s.row_idx >> (l_skip + v.n_lift);
                EF e = EF::one();
                for (unsigned j = l_skip + v.n_lift; j < l_skip + n_logup; j++) {
                    e = e * eq1b(xi[j], b_int & 1);
                    b_int >>= 1;
                }
                tp[t].eq3b.push_back(e);
            }
            HostTime ht_trace("round0.per_trace_setup");
            {
                tp[t].zc_w = lambda_pows;
                tp[t].zc = std::make_unique<ProgDev>(make_prog(std::move(progs[t].zc), lambda_pows, true, v.log_lifted, 1));
                if (v.n_lift > 0)
                    tp[t].zc_g = make_grouped(std::move(progs[t].zc_g));
            }
            if (!a.interactions.empty()) {
                std::vector<EF> wts;  // in the order roots_logup numbers them
                for (size_t k = 0; k < a.interactions.size(); k++) {
                    const Interaction &it = a.interactions[k];
                    for (size_t j = 0; j < it.message.size(); j++) wts.push_back(tp[t].eq3b[k] * beta_pows[j]);
                    wts.push_back(tp[t].eq3b[k]);
                    tp[t].lg_const += tp[t].eq3b[k] * (beta_pows[it.message.size()] * fp((uint64_t)it.bus_index + 1));
                }
                tp[t].lg_w = wts;
                tp[t].lg = std::make_unique<ProgDev>(make_prog(std::move(progs[t].lg), wts, true, v.log_lifted, 2));
                if (v.n_lift > 0)
                    tp[t].lg_g = make_grouped(std::move(progs[t].lg_g));
            }
            job.zc_cosets = cdeg ? cdeg - 1 : 0;
            job.lg_cosets = tp[t].lg ? cdeg : 0;
            const unsigned n_cosets = std::max(job.zc_cosets, job.lg_cosets);
            if (n_cosets > MAX_COSETS) throw OvmFail(801, "constraint degree too large");
            if (!n_cosets) continue;
            // eq(xi[l_skip..l_skip+n_lift], x)
            job.d_eq.alloc(size_t(1) << v.n_lift);
            OVM_THROW_IF(eq_table(d_xi.p + l_skip, v.n_lift, job.d_eq.p, cx.stream));
            EarlyLde &early_t = early.per_trace[t];
            const bool lde_done = early_t.done && early_t.n_cosets == n_cosets;
            const std::vector<uint32_t> used =
                lde_done ? early_t.used : used_columns(t, tp[t].zc->prog.used_cols, tp[t].lg ? &tp[t].lg->prog.used_cols : nullptr);
            if (lde_done) job.d_used = std::move(early_t.d_used);
            else job.d_used = cx.upload(used);
            // zerocheck_save_memory: a tall trace is extended and evaluated in slabs of 2^slab rows -- the LDE buffer
            // (n_cosets x columns x rows words: 3.5 GB at config D, 14 GB at config E) shrinks to one slab; the slab
            // sums are linear and are added up on the host. (The chunks of 2^l_skip rows are independent.)
            const unsigned slab_log = slab_logs[t];
            job.n_slabs = 1u << (v.log_lifted - slab_log);
            if (lde_done) job.lde = std::move(early_t.lde);
            else job.lde.alloc((size_t)n_cosets * v.cm.total << slab_log);
            const size_t red_blocks = std::min<size_t>(RED_BLOCKS, ((size_t(1) << slab_log) + RED_THREADS - 1) / RED_THREADS);
            const bool batch_zc = job.zc_cosets && job.n_slabs == 1 && r0_batchable(tp[t].zc->staged, v.log_lifted);
            const bool batch_lg = job.lg_cosets && job.n_slabs == 1 && r0_batchable(tp[t].lg->staged, v.log_lifted);
            if (job.zc_cosets) {
                if (batch_zc && tp[t].zc_g && tp[t].zc_g->gp.off.size() - 1 >= r0_groups_min)
                    tp[t].zc_sg = make_staged_groups(tp[t].zc_g->gp, tp[t].zc_w, v.log_lifted, 1,
                                                     R0_GROUP_JOB_WGS / round0_staged_grid(tp[t].zc->staged, v.log_lifted));
                const size_t zc_records = std::max<size_t>(red_blocks, batch_zc ? r0_job_grid(tp[t].zc->staged, tp[t].zc_sg.get(), v.log_lifted) : 0);
                job.zc_partials.alloc(zc_records * job.zc_cosets * N + 64);
                job.zc_slot = next_slot;
                job.zc_stride = (job.zc_cosets * N + 63) & ~size_t(63);
                next_slot += job.zc_stride * job.n_slabs;
            }
            if (job.lg_cosets) {
                if (batch_lg && tp[t].lg_g && tp[t].lg_g->gp.off.size() - 1 >= r0_groups_min)
                    tp[t].lg_sg = make_staged_groups(tp[t].lg_g->gp, tp[t].lg_w, v.log_lifted, 2,
                                                     R0_GROUP_JOB_WGS / round0_staged_grid(tp[t].lg->staged, v.log_lifted));
                const size_t lg_records = std::max<size_t>(red_blocks, batch_lg ? r0_job_grid(tp[t].lg->staged, tp[t].lg_sg.get(), v.log_lifted) : 0);
                job.lg_partials.alloc(lg_records * job.lg_cosets * 2 * N + 64);
                job.lg_slot = next_slot;
                job.lg_stride = (job.lg_cosets * 2 * N + 63) & ~size_t(63);
                next_slot += job.lg_stride * job.n_slabs;
            }
            if (next_slot + 4 * MAX_COSETS * N + 128 > Mailbox::SLOTS) throw OvmFail(801, "too many traces for the round-0 mailbox");
            for (uint32_t sb = 0; sb < job.n_slabs; sb++) {
                const uint64_t row0 = (uint64_t)sb << slab_log;
                const EF *eq_slab = job.d_eq.p + (row0 >> l_skip);
                if (!lde_done) {
                    OVM_PROF("round0.coset_lde", cx.stream);
                    OVM_THROW_IF(coset_lde(v.cols.p, job.d_used.p, (uint32_t)used.size(), (uint32_t)v.cm.total, slab_log, l_skip,
                                           n_cosets, d_sp_all.p, job.lde.p, cx.stream, row0));
                }
                if (job.zc_cosets) {
                    OVM_PROF("round0.zerocheck", cx.stream);
                    const size_t slot = job.zc_slot + sb * job.zc_stride;
                    if (!(batch_zc && r0_enqueue(tp[t].zc->staged, tp[t].zc_sg.get(), job.lde.p, (uint32_t)v.cm.total, v.log_lifted, job.zc_cosets, 1,
                                                 job.d_eq.p, EF::zero(), job.zc_partials.p, slot)))
                        OVM_THROW_IF(round0_eval(tp[t].zc->dev(), tp[t].zc->staged, job.lde.p, (uint32_t)v.cm.total, slab_log, l_skip,
                                                 job.zc_cosets, 1, eq_slab, EF::zero(), mb.arm(job.zc_partials.p, slot), cx.stream));
                }
                if (job.lg_cosets) {
                    OVM_PROF("round0.logup", cx.stream);
                    const size_t slot = job.lg_slot + sb * job.lg_stride;
                    if (!(batch_lg && r0_enqueue(tp[t].lg->staged, tp[t].lg_sg.get(), job.lde.p, (uint32_t)v.cm.total, v.log_lifted, job.lg_cosets, 2,
                                                 job.d_eq.p, tp[t].lg_const, job.lg_partials.p, slot)))
                        OVM_THROW_IF(round0_eval(tp[t].lg->dev(), tp[t].lg->staged, job.lde.p, (uint32_t)v.cm.total, slab_log, l_skip,
                                                 job.lg_cosets, 2, eq_slab, tp[t].lg_const, mb.arm(job.lg_partials.p, slot), cx.stream));
                }
            }
            if (next_slot + 4 * MAX_COSETS * N + 128 > Mailbox::SLOTS) throw OvmFail(801, "too many traces for the round-0 mailbox");
        }
        DevBuf<uint32_t> r0_counters(2 * T + 8);
        {
            size_t n_batched = 0;
            for (auto &byrows : r0_groups)
                for (auto &g : byrows) n_batched += g.size();
            if (n_batched) {
                OVM_PROF("round0.batched", cx.stream);
                OVM_THROW_IF(hipMemsetAsync(r0_counters.p, 0, (2 * T + 8) * sizeof(uint32_t), cx.stream));
                size_t counter_off = 0;
                for (int lr = 0; lr < 3; lr++)
                    for (int no = 0; no < 2; no++) {
                        std::vector<R0Job> &g = r0_groups[lr][no];
                        if (g.empty()) continue;
                        uint32_t wg = 0;
                        for (auto &j : g) {
                            j.wg0 = wg;
                            wg += j.gx;
                        }
                        auto d_jobs = cx.upload(g);
                        OVM_THROW_IF(round0_eval_batched(d_jobs.p, g.data(), (uint32_t)g.size(), 1u << lr, (uint32_t)no + 1, l_skip,
                                                         r0_counters.p + counter_off, mb.arm(nullptr, 0), cx.stream));
                        counter_off += g.size() + 1;
                    }
            }
        }
        if (next_slot) mb.wait(cx.stream);
        for (size_t t = 0; t < T; t++) {
            if (!mine(t)) continue;
            Round0Job &job = jobs[t];
            TraceView &v = tv[t];
            const unsigned cdeg = job.cdeg;
            if (job.zc_cosets) {
                auto ev = mb.read(job.zc_slot, (size_t)job.zc_cosets * N);
                for (uint32_t sb = 1; sb < job.n_slabs; sb++) {
                    const auto more = mb.read(job.zc_slot + sb * job.zc_stride, (size_t)job.zc_cosets * N);
                    for (size_t i = 0; i < ev.size(); i++) ev[i] += more[i];
                }
                std::vector<std::vector<EF>> ce(job.zc_cosets, std::vector<EF>(N));
                for (unsigned k = 0; k < job.zc_cosets; k++) {
                    Fp zinv = (g.pow(k + 1).exp_power_of_2(l_skip) - Fp::one()).inv();
                    for (size_t z = 0; z < N; z++) ce[k][z] = ev[k * N + z] * zinv;
                }
                std::vector<EF> q = from_coset_evals(ce, l_skip);
                unsigned deg = cdeg * (unsigned)(N - 1);
                std::vector<EF> coeffs(deg + 1);
                for (size_t i = 0; i <= deg; i++) {
                    EF c = i < q.size() ? -q[i] : EF::zero();
                    if (i >= N && i - N < q.size()) c += q[i - N];
                    coeffs[i] = c;
                }
                sp0[2 * T + t] = std::move(coeffs);
            } else if (cdeg != 0 && !job.lg_cosets) {
                sp0[2 * T + t] = std::vector<EF>(cdeg * (N - 1) + 1, EF::zero());
            }
            if (job.lg_cosets) {
                auto ev = mb.read(job.lg_slot, (size_t)job.lg_cosets * 2 * N);
                for (uint32_t sb = 1; sb < job.n_slabs; sb++) {
                    const auto more = mb.read(job.lg_slot + sb * job.lg_stride, (size_t)job.lg_cosets * 2 * N);
                    for (size_t i = 0; i < ev.size(); i++) ev[i] += more[i];
                }
                std::vector<std::vector<EF>> cn(job.lg_cosets, std::vector<EF>(N)), cden(job.lg_cosets, std::vector<EF>(N));
                for (unsigned k = 0; k < job.lg_cosets; k++)
                    for (size_t z = 0; z < N; z++) {
                        cn[k][z] = ev[(k * 2 + 0) * N + z];
                        cden[k][z] = ev[(k * 2 + 1) * N + z];
                    }
                Fp norm = fp(uint64_t(1) << (v.log_lifted - v.log_height)).inv();
                std::vector<EF> pn = from_coset_evals(cn, l_skip), pd = from_coset_evals(cden, l_skip);
                for (auto &x : pn) x = x * norm;
                sp0[2 * t] = std::move(pn);
                sp0[2 * t + 1] = std::move(pd);
            }
            job = Round0Job();  // release the LDE and scratch buffers
        }

        // (sharded: the coefficient vectors of a trace come from its owner)
        share_vectors(sp0, [&](size_t i) { return owner[i < 2 * T ? i / 2 : i - 2 * T]; });
        wt.mark("rap.round0");
        if (wt.on) HostTime::report();
        // ---- s_0 assembly (host) ----
        const unsigned s_deg = cdeg_global + 1;
        S0 s0r = assemble_s0(sp0, xi);
        const std::vector<EF> &numer_terms = s0r.numer_terms, &denom_terms = s0r.denom_terms, &mu_pows = s0r.mu_pows, &s0 = s0r.s0;
        std::vector<EF> r;
        EF r0 = ts.sample_ext();
        r.push_back(r0);
        EF prev_s = horner(s0, r0);

        wt.mark("rap.s0_host");
        // ---- fold at r_0 ----
        {
            // barycentric weights: L_z(r) = omega^z (r^N - 1) / (N (r - omega^z))
            EF rn = r0.exp_power_of_2(l_skip) - Fp::one();
            Fp ninv = fp(N).inv();
            std::vector<EF> wts(N);
            for (size_t z = 0; z < N; z++) wts[z] = rn * (r0 - omega_pows[z]).inv() * (omega_pows[z] * ninv);
            auto d_w = cx.upload(wts);
            OVM_PROF("constraints.fold_ple", cx.stream);
            for (size_t t = 0; t < T; t++) {
                if (!mine(t)) continue;
                TraceView &v = tv[t];
                size_t h = size_t(1) << v.n_lift;
                v.mats_a.alloc(v.cm.total * h);
                v.mats_b.alloc(v.cm.total * std::max<size_t>(h / 2, 1));
                OVM_THROW_IF(fold_ple(v.cols.p, v.twins.p, (uint32_t)v.cm.total, v.log_lifted, l_skip, d_w.p, v.mats_a.p, cx.stream));
                v.cur_log_h = v.n_lift;
                v.mats_in_a = true;
                if (v.n_lift == 0) v.final_row = cx.download(v.mats_a.p, v.cm.total);
            }
        }
        std::vector<EF> eq_ns{eq_uni(l_skip, xi[0], r0)};
        EF eq_sharp0 = EF::zero();
        {
            std::vector<EF> eqx = eq_hypercube(std::vector<EF>(xi.begin(), xi.begin() + l_skip));
            for (size_t z = 0; z < N; z++) eq_sharp0 += eq_uni(l_skip, r0, ef(omega_pows[z])) * eqx[z];
        }
        std::vector<EF> eq_sharp_ns{eq_sharp0};

        wt.mark("rap.fold_ple");
        // ---- MLE rounds ----
        std::vector<EF> zc_tilde(T, EF::zero());
        std::vector<std::array<EF, 2>> lg_tilde(T, {EF::zero(), EF::zero()});
        std::vector<std::vector<EF>> round_polys;
        // one eq table per distinct trace height and round (shared by all traces of that height);
        // per trace and program its own partial-sum scratch and 8 mailbox slots
        std::vector<DevBuf<EF>> d_eqr_by_nlift(n_max + 1);
        std::vector<DevBuf<EF>> mle_partials(2 * T);
        for (size_t t = 0; t < T; t++) {
            if (tv[t].n_lift == 0 || !mine(t)) continue;
            if (!d_eqr_by_nlift[tv[t].n_lift].p) d_eqr_by_nlift[tv[t].n_lift].alloc(size_t(1) << tv[t].n_lift);
            // records of 8 sums per (workgroup, group); the batched kernel's lanes enumerate (pair, point), so it can
            // run up to `degree` times the workgroups of the per-pair kernels
            const size_t wgs = std::min<size_t>(RED_BLOCKS, (((size_t)cdeg_global << (tv[t].n_lift - 1)) + RED_THREADS - 1) / RED_THREADS);
            // (every launcher keeps workgroups x groups within RED_BLOCKS records, or one workgroup per group)
            auto records = [&](size_t groups) { return std::min(wgs * groups, std::max<size_t>(RED_BLOCKS, groups)); };
            mle_partials[2 * t].alloc(records(tp[t].zc_g->gp.off.size() - 1) * 8 + 64);
            if (!air(t).interactions.empty()) mle_partials[2 * t + 1].alloc(records(tp[t].lg_g->gp.off.size() - 1) * 8 + 64);
        }
        if (16 * T + 64 > Mailbox::SLOTS) throw OvmFail(801, "too many traces for the MLE-round mailbox");
        // Jobs of at most MLE_BATCH_MAX_WGS workgroups go into one launch per round (the late rounds
        // of a many-AIR proof are otherwise dozens of tiny launches serialised by the stream).
        // (OVM_MLE_BATCH=0 launches every job on its own: parity tests of the unbatched kernels on small traces)
        const uint32_t MLE_BATCH_MAX_WGS = tuning().mle_batch_wgs;  // (0: no batching)
        // (two batches per round: the (pair, point)-per-lane kernel for the short jobs and the LDS-slot kernel for jobs of at
        // least 2^mle_batch_lds_min_log pairs; each with its own arrival counters)
        DevBuf<uint32_t> mle_counters(2 * (2 * T + 1));
        OVM_THROW_IF(hipMemsetAsync(mle_counters.p, 0, 2 * (2 * T + 1) * sizeof(uint32_t), cx.stream));
        const uint32_t batch_lds_min_log = tuning().mle_batch_lds_min_log;
        OVM_PROF("constraints.mle_rounds", cx.stream);
        for (unsigned round = 1; round <= n_max; round++) {
            EF r_prev = r[round - 1];
            std::vector<std::vector<EF>> sp(3 * T);
            // phase A: enqueue the round sums of every trace that still has rows to fold
            std::vector<char> eq_done(n_max + 1, 0);
            bool any_launch = false;
            std::vector<MleJob> jobs, jobs_lds;
            auto enqueue = [&](const GroupedDev &gd, const EF *weights, const EF *mats, unsigned log_h, uint32_t n_out,
                               const EF *eq, EF add_const, EF *partials, size_t partials_cap, size_t slot) {
                const uint32_t n_groups = (uint32_t)gd.gp.off.size() - 1;
                const bool lds = log_h > batch_lds_min_log && mle_lds_eligible(gd.gp.max_slots, cdeg_global);
                const uint32_t gx = lds ? mle_round_gx(log_h, cdeg_global, n_groups, true) : mle_round_batched_gx(log_h, cdeg_global, n_groups);
                if (gx * n_groups <= MLE_BATCH_MAX_WGS) {
                    std::vector<MleJob> &into = lds ? jobs_lds : jobs;
                    MleJob j{};
                    j.code = gd.code.p, j.off = gd.off.p, j.weights = weights, j.mats = mats, j.eq = eq;
                    j.partials = partials, j.out = mb.slot_ptr(slot), j.add_const = add_const;
                    j.log_h = log_h, j.n_out = n_out, j.n_groups = n_groups, j.gx = gx;
                    j.z0 = into.empty() ? 0 : into.back().z0 + into.back().n_groups;
                    j.max_slots = gd.gp.max_slots;
                    j.partials_cap = (uint32_t)partials_cap;
                    into.push_back(j);
                } else {
                    OVM_THROW_IF(mle_round_eval_grouped(gd.code.p, gd.off.p, n_groups, gd.gp.max_slots, weights, mats, log_h,
                                                        cdeg_global, n_out, eq, add_const, mb.arm(partials, slot), cx.stream));
                }
            };
            for (size_t t = 0; t < T; t++) {
                TraceView &v = tv[t];
                if (round > v.n_lift || !mine(t)) continue;
                const bool has_int = !air(t).interactions.empty();
                const unsigned log_num_y = v.n_lift - round;
                DevBuf<EF> &d_eqr = d_eqr_by_nlift[v.n_lift];
                if (!eq_done[v.n_lift]) {
                    // eq over xi[l_skip + round .. l_skip + n_lift]
                    OVM_THROW_IF(eq_table(d_xi.p + l_skip + round, log_num_y, d_eqr.p, cx.stream));
                    eq_done[v.n_lift] = 1;
                }
                const EF *mats = v.mats_in_a ? v.mats_a.p : v.mats_b.p;
                enqueue(*tp[t].zc_g, tp[t].zc->weights.p, mats, v.cur_log_h, 1, d_eqr.p, EF::zero(), mle_partials[2 * t].p, mle_partials[2 * t].n, 16 * t);
                if (has_int)
                    enqueue(*tp[t].lg_g, tp[t].lg->weights.p, mats, v.cur_log_h, 2, d_eqr.p, tp[t].lg_const,
                            mle_partials[2 * t + 1].p, mle_partials[2 * t + 1].n, 16 * t + 8);
                any_launch = true;
            }
            if (!jobs_lds.empty()) {
                auto d_jobs = cx.upload(jobs_lds);
                OVM_THROW_IF(mle_round_eval_batched_lds(d_jobs.p, jobs_lds.data(), (uint32_t)jobs_lds.size(), cdeg_global,
                                                        mle_counters.p + (2 * T + 1), mb.arm(nullptr, 0), cx.stream));
            }
            if (!jobs.empty()) {  // armed last: its flag means every reduction of the round is done
                auto d_jobs = cx.upload(jobs);
                OVM_THROW_IF(mle_round_eval_batched(d_jobs.p, jobs.data(), (uint32_t)jobs.size(), cdeg_global, mle_counters.p,
                                                    mb.arm(nullptr, 0), cx.stream));
            }
            // While the round's kernels run: the traces that have just run out of rows (their last row came back with the
            // previous round's fold) are evaluated on the host -- the whole constraint program on one row of extension-field
            // values, 50-500 us for the wide AIRs of a reth segment, which used to happen after the wait, device idle.
            for (size_t t = 0; t < T; t++) {
                if (!mine(t) || round != tv[t].n_lift + 1) continue;
                TraceView &v = tv[t];
                const bool has_int = !air(t).interactions.empty();
                const Fp norm = fp(uint64_t(1) << (v.log_lifted - v.log_height)).inv();
                EF o[2];
                host_eval(tp[t].zc->prog, tp[t].zc_w, v.final_row, EF::zero(), o);
                zc_tilde[t] = eq_ns.back() * o[0];
                if (has_int) {
                    host_eval(tp[t].lg->prog, tp[t].lg_w, v.final_row, tp[t].lg_const, o);
                    lg_tilde[t] = {eq_sharp_ns.back() * o[0] * norm, eq_sharp_ns.back() * o[1]};
                }
            }
            if (any_launch) mb.wait(cx.stream);
            // phase B: host (the owner of a trace evaluates its round polynomials; sharded: then all-gathered)
            for (size_t t = 0; t < T; t++) {
                if (!mine(t)) continue;
                TraceView &v = tv[t];
                const AirKey &a = air(t);
                bool has_int = !a.interactions.empty();
                Fp norm = fp(uint64_t(1) << (v.log_lifted - v.log_height)).inv();
                if (round > v.n_lift) {
                    if (round == v.n_lift + 1) {
                        // (evaluated above, while the device was busy)
                    } else {
                        zc_tilde[t] *= r_prev;
                        lg_tilde[t][0] *= r_prev;
                        lg_tilde[t][1] *= r_prev;
                    }
                    sp[2 * T + t] = {zc_tilde[t]};
                    sp[2 * t] = has_int ? std::vector<EF>{lg_tilde[t][0]} : std::vector<EF>(cdeg_global, EF::zero());
                    sp[2 * t + 1] = has_int ? std::vector<EF>{lg_tilde[t][1]} : std::vector<EF>(cdeg_global, EF::zero());
                } else {
                    auto ev = mb.read(16 * t, 8);
                    sp[2 * T + t].resize(cdeg_global);
                    for (unsigned i = 0; i < cdeg_global; i++) sp[2 * T + t][i] = ev[2 * i];
                    if (has_int) {
                        auto e2 = mb.read(16 * t + 8, 8);
                        sp[2 * t].resize(cdeg_global);
                        sp[2 * t + 1].resize(cdeg_global);
                        for (unsigned i = 0; i < cdeg_global; i++) {
                            sp[2 * t][i] = e2[2 * i] * norm;
                            sp[2 * t + 1][i] = e2[2 * i + 1];
                        }
                    } else {
                        sp[2 * t] = std::vector<EF>(cdeg_global, EF::zero());
                        sp[2 * t + 1] = std::vector<EF>(cdeg_global, EF::zero());
                    }
                }
            }
            share_vectors(sp, [&](size_t i) { return owner[i < 2 * T ? i / 2 : i - 2 * T]; });
            size_t tail_start = T;
            for (size_t t = 0; t < T; t++)
                if ((int)round > tv[t].n) {
                    tail_start = t;
                    break;
                }
            std::vector<EF> head_zc(cdeg_global, EF::zero()), head_lg(cdeg_global, EF::zero());
            EF sp_tail = EF::zero();
            for (size_t t = 0; t < T; t++) {
                size_t zi = 2 * T + t, ni = 2 * t, di = 2 * t + 1;
                if (t < tail_start) {
                    for (unsigned i = 0; i < cdeg_global; i++) {
                        head_zc[i] += mu_pows[zi] * sp[zi][i];
                        head_lg[i] += mu_pows[ni] * sp[ni][i] + mu_pows[di] * sp[di][i];
                    }
                } else {
                    sp_tail += mu_pows[zi] * sp[zi][0] + mu_pows[ni] * sp[ni][0] + mu_pows[di] * sp[di][0];
                }
            }
            std::vector<EF> head(s_deg, EF::zero());
            for (unsigned i = 0; i < cdeg_global; i++) head[i + 1] = eq_ns[round - 1] * head_zc[i] + eq_sharp_ns[round - 1] * head_lg[i];
            EF xi_cur = xi[l_skip + round - 1];
            head[0] = (prev_s - xi_cur * head[1] - sp_tail) * (EF::one() - xi_cur).inv();
            std::vector<Fp> pts(s_deg);
            for (unsigned i = 0; i < s_deg; i++) pts[i] = fp(i);
            std::vector<EF> coeffs = lagrange_coeffs(pts, head);
            coeffs.push_back(EF::zero());
            EF b = EF::one() - xi_cur, a2 = xi_cur - b;
            for (size_t i = s_deg; i-- > 0;) coeffs[i + 1] = a2 * coeffs[i] + b * coeffs[i + 1];
            coeffs[0] = coeffs[0] * b;
            coeffs[1] += sp_tail;
            std::vector<EF> be(s_deg);
            for (unsigned i = 1; i <= s_deg; i++) {
                be[i - 1] = horner(coeffs, ef_u64(i));
                ts.observe_ext(be[i - 1]);
            }
            round_polys.push_back(be);
            EF rr = ts.sample_ext();
            r.push_back(rr);
            prev_s = horner(coeffs, rr);
            {
                // matrices of at most 2^11 rows fold in one launch per round (as many as fit its grid)
                std::vector<FoldJob> fj;
                auto flush_folds = [&] {
                    if (fj.empty()) return;
                    if (fj.size() <= FoldJobsArg::MAX) {
                        OVM_THROW_IF(fold_mle_batched(nullptr, fj.data(), (uint32_t)fj.size(), rr, cx.stream));
                    } else {
                        auto d_fj = cx.upload(fj);
                        OVM_THROW_IF(fold_mle_batched(d_fj.p, fj.data(), (uint32_t)fj.size(), rr, cx.stream));
                    }
                    fj.clear();
                };
                for (size_t t = 0; t < T; t++) {
                    TraceView &v = tv[t];
                    if (v.cur_log_h == 0 || v.cm.total == 0) continue;
                    const EF *in = v.mats_in_a ? v.mats_a.p : v.mats_b.p;
                    EF *o = v.mats_in_a ? v.mats_b.p : v.mats_a.p;
                    if (v.cur_log_h <= 11 && v.cm.total <= 32768) {
                        const uint32_t y0 = fj.empty() ? 0 : fj.back().y0 + fj.back().n_cols;
                        if (y0 + v.cm.total > 65535) flush_folds();
                        fj.push_back(FoldJob{in, o, (uint32_t)v.cm.total, v.cur_log_h, fj.empty() ? 0 : fj.back().y0 + fj.back().n_cols, 0});
                    } else {
                        OVM_THROW_IF(fold_mle(in, o, (uint32_t)v.cm.total, v.cur_log_h, rr, cx.stream));
                    }
                }
                flush_folds();
                for (size_t t = 0; t < T; t++) {
                    TraceView &v = tv[t];
                    if (v.cur_log_h == 0) continue;
                    v.mats_in_a = !v.mats_in_a;
                    v.cur_log_h--;
                    if (v.cur_log_h == 0) v.final_row = cx.download(v.mats_in_a ? v.mats_a.p : v.mats_b.p, v.cm.total);
                }
            }
            EF eqr = eq1(xi_cur, rr);
            eq_ns.push_back(eq_ns[round - 1] * eqr);
            eq_sharp_ns.push_back(eq_sharp_ns[round - 1] * eqr);
        }

        wt.mark("rap.mle_rounds");
        // ---- openings ---- (cpu.rs:644-694, mod.rs:408-424): per trace the parts [common main,
        // preprocessed?, cached_0, ...], each as (claim, rotated claim) pairs when the AIR needs rotation
        if (!sh.single()) {
            std::vector<std::vector<EF>> rows(T);
            for (size_t t = 0; t < T; t++)
                if (mine(t)) rows[t] = tv[t].final_row;
            share_vectors(rows, [&](size_t t) { return owner[t]; });
            for (size_t t = 0; t < T; t++) {
                if (rows[t].size() != tv[t].cm.total) throw OvmFail(1, "shard exchange: column openings of the wrong length");
                tv[t].final_row = std::move(rows[t]);
            }
        }
        std::vector<std::vector<std::vector<EF>>> openings(T);
        for (size_t t = 0; t < T; t++) {
            TraceView &v = tv[t];
            auto part = [&](size_t wd, uint32_t base) {
                std::vector<EF> o;
                for (size_t c = 0; c < wd; c++) {
                    o.push_back(v.final_row[base + c]);
                    if (v.cm.need_rot) o.push_back(v.final_row[base + wd + c]);
                }
                return o;
            };
            const size_t n_cached = air(t).cached_widths.size();
            openings[t].push_back(part(air(t).common_main_width, v.cm.col((uint32_t)n_cached, 0, 0)));
            if (v.cm.pre_width) openings[t].push_back(part(v.cm.pre_width, v.cm.pre_col(0, 0)));
            for (size_t k = 0; k < n_cached; k++)
                openings[t].push_back(part(air(t).cached_widths[k], v.cm.col((uint32_t)k, 0, 0)));
        }
        auto observe_part = [&](const std::vector<EF> &o, bool rot) {
            for (auto &x : o) {
                ts.observe_ext(x);
                if (!rot) ts.observe_ext(EF::zero());
            }
        };
        for (size_t t = 0; t < T; t++) observe_part(openings[t][0], tv[t].cm.need_rot);  // common mains first
        for (size_t t = 0; t < T; t++)
            for (size_t k = 1; k < openings[t].size(); k++) observe_part(openings[t][k], tv[t].cm.need_rot);

        // batch constraint proof bytes
        w.e_slice(numer_terms);
        for (auto &x : denom_terms) w.e(x);
        w.e_slice(s0);
        w.usize(round_polys.size());
        if (!round_polys.empty()) {
            w.usize(round_polys[0].size());
            for (auto &rp : round_polys)
                for (auto &x : rp) w.e(x);
        }
        for (size_t t = 0; t < T; t++) {
            w.usize(openings[t].size());
            for (auto &o : openings[t]) w.e_slice(o);
        }
        section_offsets.push_back(w.buf.size());
        return r;
    }

    // -----------------------------------------------------------------------------------------
    // Stage 3: stacked opening reduction + WHIR (OpeningProver::prove_openings)
    // -----------------------------------------------------------------------------------------
    // The commitments of a proof: the common main first, then per trace (sorted order) its preprocessed
    // commitment (if any) and its cached mains (cpu-backend/src/device.rs:147-172). Layout, root and owner
    // are known on every rank; the data (`pcs`) only on the owner -- rank 0 for the common main, the owner
    // of the trace for the others.
    struct CommitRef {
        const PcsData *pcs = nullptr;
        Layout layout;
        p2::Digest root;
        uint32_t owner = 0;
        std::vector<bool> need_rot;  // per stacked matrix
    };
    Layout common_main_layout() const {
        std::vector<std::pair<size_t, unsigned>> meta;
        for (size_t t = 0; t < traces.size(); t++) meta.push_back({air(t).common_main_width, log2_strict(traces[t].height)});
        return Layout(P().l_skip, P().l_skip + P().n_stack, meta);
    }
    std::vector<CommitRef> gather_commits(const PcsData *common, const p2::Digest &common_root) const {
        const unsigned l_skip = P().l_skip, lh = l_skip + P().n_stack;
        std::vector<CommitRef> commits(1);
        commits[0].pcs = common;
        commits[0].layout = common_main_layout();
        commits[0].root = common_root;
        commits[0].owner = 0;
        for (size_t t = 0; t < traces.size(); t++) commits[0].need_rot.push_back(air(t).need_rot);
        for (size_t t = 0; t < traces.size(); t++) {
            const unsigned log_h = log2_strict(traces[t].height);
            if (air(t).preprocessed_width) {
                if (!pre_of(t)) throw OvmFail(1, "preprocessed commitment missing");
                CommitRef c;
                c.pcs = mine(t) ? pre_of(t) : nullptr;
                c.layout = Layout(l_skip, lh, {{air(t).preprocessed_width, log_h}});
                c.root = pre_of(t)->root;
                c.owner = owner[t];
                c.need_rot = {air(t).need_rot};
                commits.push_back(std::move(c));
            }
            if (traces[t].cached_roots.size() != air(t).cached_widths.size()) throw OvmFail(1, "cached main partitions: wrong count");
            for (size_t k = 0; k < air(t).cached_widths.size(); k++) {
                CommitRef c;
                c.pcs = mine(t) ? traces[t].cached[k] : nullptr;
                c.layout = Layout(l_skip, lh, {{air(t).cached_widths[k], log_h}});
                c.root = traces[t].cached_roots[k];
                c.owner = owner[t];
                c.need_rot = {air(t).need_rot};
                commits.push_back(std::move(c));
            }
        }
        for (auto &c : commits)
            if (c.owner == sh.rank && (!c.pcs || c.pcs->log_height != lh || c.pcs->layout.width != c.layout.width))
                throw OvmFail(1, "commitment missing or of a different stacked shape");
        return commits;
    }
    // v += the same vector of every other rank (sharded proving; identity on one rank)
    void share_sum(std::vector<EF> &v) {
        if (sh.single() || v.empty()) return;
        std::vector<EF> all(v.size() * sh.world);
        sh.allgather(v.data(), all.data(), v.size() * sizeof(EF));
        for (size_t i = 0; i < v.size(); i++) {
            EF acc = EF::zero();
            for (uint32_t r = 0; r < sh.world; r++) acc += all[r * v.size() + i];
            v[i] = acc;
        }
    }

    // Opened rows of one initial commitment at WHIR round-0 query indices (whir.rs: the rows {i + t * query_stride} of
    // query i; physical order, see merkle.hip): vals[(q * 2^k + t) * width + c].
    void open_rows(const PcsData &pcs, const std::vector<uint64_t> &qidx, unsigned lq, std::vector<Fp> &vals) {
        const unsigned k_whir = P().k_whir;
        std::vector<uint64_t> rows;
        for (uint64_t i : qidx) {
            uint64_t grp = bitrev64(i, lq);
            for (uint64_t t = 0; t < (1ull << k_whir); t++) rows.push_back((grp << k_whir) + bitrev64(t, k_whir));
        }
        const size_t cw = pcs.layout.width;
        auto d_rows = cx.upload(rows);
        DevBuf<Fp> d_vals(std::max<size_t>(rows.size() * cw, 1));
        const Fp *codeword = pcs.codeword.p;
        DevBuf<Fp> recomputed;
        if (!codeword) {  // cache_rs_code_matrix = false at commit time: encode again (one NTT of the commitment)
            OVM_PROF("whir.reencode", cx.stream);
            recomputed.alloc(cw << pcs.log_rows);
            OVM_THROW_IF(ovm_hip_ntt_dif((const uint32_t *)pcs.msg.p, (uint32_t *)recomputed.p, cw, pcs.log_rows, pcs.log_height, cx.stream));
            codeword = recomputed.p;
        }
        OVM_THROW_IF(ovm_hip_gather_rows((const uint32_t *)codeword, cw, 1ull << pcs.log_rows, 1, d_rows.p, rows.size(),
                                         (uint32_t *)d_vals.p, cx.stream));
        vals = cx.download(d_vals.p, rows.size() * cw);
    }
    // Where the Merkle-path siblings of the queries live, in proof order (query-major, bottom layer first): the
    // sibling of query i at stored layer j is entry (i >> j) ^ 1, at physical position bitrev of it (merkle.hip).
    // A tree built by one rank holds them all; in a tree built by dist_n ranks together (build_tree) the entry at
    // physical position p of a layer with more than dist_n entries is in rank p mod dist_n's share at compact
    // position p / dist_n, and the layers of at most dist_n entries are the host copy `top`.
    struct PathRef {
        int rank;      // -1: pcs.top[idx] (host, every rank); else: layers[idx] on that rank
        uint64_t idx;
    };
    std::vector<PathRef> path_refs(const PcsData &pcs, int owner_rank, const std::vector<uint64_t> &qidx, unsigned lq) const {
        const uint32_t n = pcs.dist_n;
        const uint64_t qs_share = (1ull << lq) / n;
        std::vector<PathRef> refs;
        for (uint64_t i : qidx)
            for (unsigned j = 0; j < lq; j++) {
                const uint64_t size = 1ull << (lq - j), p = bitrev64((i >> j) ^ 1, lq - j);
                if (n > 1 && size <= n) {
                    refs.push_back({-1, (2ull * n - 2 * size) + p});
                } else {
                    uint64_t off = 0;
                    for (unsigned m2 = 0; m2 < j; m2++) off += qs_share >> m2;
                    refs.push_back({n > 1 ? (int)(p % n) : owner_rank, off + p / n});
                }
            }
        return refs;
    }
    // the digests of `refs` that live in THIS rank's layers, in order
    std::vector<p2::Digest> serve_paths(const PcsData &pcs, const std::vector<PathRef> &refs) {
        std::vector<uint64_t> idx;
        for (auto &r : refs)
            if (r.rank == (int)sh.rank) idx.push_back(r.idx);
        if (idx.empty()) return {};
        auto d_idx = cx.upload(idx);
        DevBuf<p2::Digest> d_dig(idx.size());
        OVM_THROW_IF(ovm_hip_gather_digests((const uint32_t *)pcs.layers.p, d_idx.p, idx.size(), (uint32_t *)d_dig.p, cx.stream));
        return cx.download(d_dig.p, idx.size());
    }

    // OpeningProver::prove_openings = the stacked opening reduction, then WHIR (the reference has the same two
    // steps: prove_stacked_opening_reduction, prover/stacked_reduction.rs:65-128, and prove_whir_opening,
    // prover/whir.rs:78-341). `common_main`: the common-main PCS data (sharded: on rank 0 only, null elsewhere);
    // its root is known everywhere.
    void prove_openings(const PcsData *common_main, const p2::Digest &common_root, const std::vector<EF> &r) {
        const std::vector<CommitRef> commits = gather_commits(common_main, common_root);
        // the stacked matrices of all commitments side by side: commitment c owns columns [coff[c], coff[c+1])
        std::vector<size_t> coff{0};
        for (auto &c : commits) coff.push_back(coff.back() + c.layout.width);
        const std::vector<EF> u = prove_stacked_reduction(commits, coff, r);
        wt.mark("openings.stacked");
        prove_whir(commits, coff, u);
    }

    // Returns the sumcheck point u (u[0] for the univariate round, then one entry per stacked round).
    std::vector<EF> prove_stacked_reduction(const std::vector<CommitRef> &commits, const std::vector<size_t> &coff_in,
                                            const std::vector<EF> &r) {
        const unsigned l_skip = P().l_skip, n_stack = P().n_stack;
        const size_t N = size_t(1) << l_skip, H = size_t(1) << (l_skip + n_stack);
        auto own = [&](size_t com) { return commits[com].owner == sh.rank; };
        // this rank holds the columns of ITS commitments, side by side, at [loff[c], loff[c] + width_c)
        const std::vector<size_t> &coff = coff_in;
        std::vector<size_t> loff(commits.size(), 0);
        size_t lwidth = 0;
        for (size_t com = 0; com < commits.size(); com++) {
            if (own(com)) {
                loff[com] = lwidth;
                lwidth += commits[com].layout.width;
            }
        }
        const size_t width = coff.back();
        ByteWriter &w = out;
        const Fp g = fp(bb::GENERATOR), omega_skip = bb::two_adic_generator(l_skip);
        std::vector<Fp> omega_pows = powers_of(omega_skip, N);

        EF lambda = ts.sample_ext();
        struct TV {
            size_t com;
            Slice s;
            size_t leq;
            long lrot;
        };
        std::vector<TV> tvs;
        size_t li = 0;
        for (size_t com = 0; com < commits.size(); com++)
            for (auto &sc : commits[com].layout.cols) {
                TV t{com, sc.slice, li, -1};
                li++;
                if (commits[com].need_rot[sc.mat_idx]) t.lrot = (long)li;
                li++;
                tvs.push_back(t);
            }
        std::vector<EF> lam = powers_of(lambda, li);
        // windows: runs of slices of one height inside one commitment (contiguous in that stacked matrix)
        std::vector<size_t> win;
        for (size_t i = 0; i < tvs.size(); i++)
            if (i == 0 || tvs[i].s.log_height != tvs[i - 1].s.log_height || tvs[i].com != tvs[i - 1].com) win.push_back(i);
        win.push_back(tvs.size());
        const size_t W = win.size() - 1;
        const EF r0 = r[0];
        const EF eq_const = eq_uni_at_one(l_skip, r0 * omega_skip);

        struct Window {
            unsigned log_height, n_lift, cur_log_h;
            int n;
            DevBuf<EF> eq_cube, vec_a, vec_b;
            bool in_a = true, own = false;
            EF eq_fin = EF::zero(), krot_fin = EF::zero();
        };
        std::vector<Window> ws(W);
        DevBuf<EF> partials((size_t)RED_BLOCKS * 3 * N + 64);
        const unsigned s0_deg = 2 * (unsigned)(N - 1);
        std::vector<EF> s0(s0_deg + 1, EF::zero());
        ProfScope *ps_st0 = new ProfScope("stacked.round0", cx.stream);
        // phase A: the round-0 sums of every window are enqueued (own partial records, own mailbox slots);
        // phase B: one wait, then the host algebra per window
        // mailbox slots of a window: its 3 N round-0 sums (64 used to be assumed, which 3 * 16 fits: with l_skip >= 5 the
        // windows of a proof with several of them overwrote each other -- found by the reference's l_skip 6 / 9 tests)
        const size_t wslots = (3 * N + 63) & ~size_t(63);
        if (wslots * W + 64 > Mailbox::SLOTS) throw OvmFail(801, "too many stacking windows for the mailbox");
        std::vector<DevBuf<EF>> r0_partials(W);
        bool any_own_window = false;
        for (size_t wi = 0; wi < W; wi++) {
            Window &wd = ws[wi];
            size_t w0 = win[wi], w1 = win[wi + 1];
            wd.log_height = tvs[w0].s.log_height;
            wd.n = (int)wd.log_height - (int)l_skip;
            wd.n_lift = (unsigned)std::max(wd.n, 0);
            wd.own = own(tvs[w0].com);
            if (!wd.own) continue;
            any_own_window = true;
            wd.eq_cube.alloc(size_t(1) << wd.n_lift);
            {
                std::vector<EF> pt(r.begin() + 1, r.begin() + 1 + wd.n_lift);
                auto d_pt = cx.upload(pt);
                OVM_THROW_IF(eq_table(d_pt.p, wd.n_lift, wd.eq_cube.p, cx.stream));
            }
            std::vector<EF> leq, lrot;
            for (size_t i = w0; i < w1; i++) {
                leq.push_back(lam[tvs[i].leq]);
                lrot.push_back(tvs[i].lrot >= 0 ? lam[(size_t)tvs[i].lrot] : EF::zero());
            }
            auto d_leq = cx.upload(leq);
            auto d_lrot = cx.upload(lrot);
            unsigned log_len = std::max(wd.log_height, l_skip);
            const Fp *base = commits[tvs[w0].com].pcs->stacked_view + tvs[w0].s.col_idx * H + tvs[w0].s.row_idx;
            bool any_rot = false;
            for (auto &x : lrot) any_rot |= !is_zero(x);
            r0_partials[wi].alloc((size_t)RED_BLOCKS * 3 * N + 64);  // (grid * slice chunks <= RED_BLOCKS records)
            OVM_THROW_IF(stacked_round0(base, (uint32_t)(w1 - w0), log_len, l_skip, d_leq.p, any_rot ? d_lrot.p : nullptr, wd.eq_cube.p,
                                        mb.arm(r0_partials[wi].p, wslots * wi), cx.stream));
        }
        if (any_own_window) mb.wait(cx.stream);
        r0_partials.clear();
        for (size_t wi = 0; wi < W; wi++) {
            Window &wd = ws[wi];
            if (!wd.own) continue;
            auto R = mb.read(wslots * wi, 3 * N);
            // host: chunk-aggregated vectors -> coset evaluations with the univariate factors
            std::vector<std::vector<EF>> coef(3);
            for (int v = 0; v < 3; v++) coef[v] = small_dft(std::vector<EF>(R.begin() + v * N, R.begin() + (v + 1) * N), true);
            std::vector<std::vector<EF>> e0(2, std::vector<EF>(N)), e1(2, std::vector<EF>(N));
            for (unsigned k = 0; k < 2; k++) {
                Fp shift = g.pow(k + 1);
                for (size_t z = 0; z < N; z++) {
                    Fp zp = shift * omega_pows[z];
                    EF lde[3];
                    for (int v = 0; v < 3; v++) lde[v] = horner_fp(coef[v], zp);
                    unsigned l = l_skip;
                    Fp omega = omega_skip;
                    EF r_uni = r0;
                    if (wd.n < 0) {
                        l = l_skip + wd.n;
                        omega = omega_skip.exp_power_of_2((unsigned)(-wd.n));
                        r_uni = r0.exp_power_of_2((unsigned)(-wd.n));
                    }
                    Fp ind = in_uni(l_skip, wd.n, zp);
                    EF equ = eq_uni(l, ef(zp), r_uni), equ_rot = eq_uni(l, ef(zp), r_uni * omega);
                    Fp equ1 = eq_uni_at_one(l_skip, zp);
                    e0[k][z] = equ * lde[0] * ind;
                    e1[k][z] = (equ_rot * lde[1] + eq_const * equ1 * lde[2]) * ind;
                }
            }
            std::vector<EF> p0 = from_coset_evals(e0, l_skip), p1 = from_coset_evals(e1, l_skip);
            for (size_t i = 0; i <= s0_deg; i++) s0[i] += p0[i] + p1[i];
        }
        share_sum(s0);  // (sharded: every rank has summed its own windows)
        delete ps_st0;
        wt.mark("openings.stacked.round0");
        for (auto &c : s0) ts.observe_ext(c);
        std::vector<EF> u;
        EF u0 = ts.sample_ext();
        u.push_back(u0);

        // fold the stacked matrix at u_0
        DevBuf<EF> q_a(std::max<size_t>(lwidth, 1) << n_stack), q_b(std::max<size_t>(lwidth, 1) << (n_stack ? n_stack - 1 : 0));
        bool q_in_a = true;
        unsigned q_log_h = n_stack;
        ProfScope *ps_stf = new ProfScope("stacked.fold_ple", cx.stream);
        {
            EF un = u0.exp_power_of_2(l_skip) - Fp::one();
            Fp ninv = fp(N).inv();
            std::vector<EF> wts(N);
            for (size_t z = 0; z < N; z++) wts[z] = un * (u0 - omega_pows[z]).inv() * (omega_pows[z] * ninv);
            auto d_w = cx.upload(wts);
            std::vector<ColDesc> cd(lwidth);
            for (size_t com = 0; com < commits.size(); com++)
                for (size_t c = 0; own(com) && c < commits[com].layout.width; c++)
                    cd[loff[com] + c] = ColDesc{commits[com].pcs->stacked_view + c * H, l_skip + n_stack, 0};
            auto d_cd = cx.upload(cd);
            if (lwidth)
                OVM_THROW_IF(fold_ple(d_cd.p, nullptr, (uint32_t)lwidth, l_skip + n_stack, l_skip, d_w.p, q_a.p, cx.stream));
            // (no synchronisation: the temporaries go back to the block cache and are only ever reused by later
            // work on this same stream)
        }
        // per-window tables and combined vectors
        std::vector<EF> tail_d(tvs.size(), EF::zero());  // D_i of the finished slices (see the round loop)
        std::vector<char> tail_ready(tvs.size(), 0);
        // The finished slices on the device (stages.hip: stacked_tail_kernel; OVM_STACKED_TAIL_DEVICE=0: the host loop):
        // their D_i are uploaded once, when their window's last fold has produced (eq_fin, krot_fin).
        // (from STACKED_TAIL_DEVICE_MIN slices on: below that the host loop is a few microseconds and a kernel per round
        // is not -- a 2^12 x 20 proof was 0.4 ms slower with its 20 slices on the device; OVM_STACKED_TAIL_DEVICE=2: always)
        static constexpr size_t STACKED_TAIL_DEVICE_MIN = 1024;
        const bool tail_on_device = !tvs.empty() && (tuning().stacked_tail_device == 2 ||
                                                     (tuning().stacked_tail_device == 1 && tvs.size() >= STACKED_TAIL_DEVICE_MIN));
        DevBuf<TailSlice> d_tail_slices;
        DevBuf<EF> d_tail, tail_partials;
        if (tail_on_device) {
            std::vector<TailSlice> ts_host(tvs.size());
            for (size_t i = 0; i < tvs.size(); i++) {
                const unsigned nl = tvs[i].s.log_height > l_skip ? tvs[i].s.log_height - l_skip : 0;
                ts_host[i] = TailSlice{(uint64_t)tvs[i].s.row_idx, own(tvs[i].com) ? (uint32_t)(loff[tvs[i].com] + tvs[i].s.col_idx) : 0u,
                                       own(tvs[i].com) ? nl : ~0u};
            }
            d_tail_slices = cx.upload(ts_host);
            d_tail.alloc(tvs.size());
            tail_partials.alloc(2 * RED_BLOCKS + 64);
        }
        auto tail_init_window = [&](size_t wi) {  // (eq_fin, krot_fin) of window wi are known: D_i of its slices
            if (!tail_on_device) return;
            const Window &wd = ws[wi];
            std::vector<EF> init;
            for (size_t i = win[wi]; i < win[wi + 1]; i++) {
                EF dd = lam[tvs[i].leq] * wd.eq_fin;
                if (tvs[i].lrot >= 0) dd += lam[(size_t)tvs[i].lrot] * wd.krot_fin;
                init.push_back(dd);
            }
            auto d_init = cx.upload(init);
            flush_uploads(cx.stream);
            OVM_THROW_IF(hipMemcpyAsync(d_tail.p + win[wi], d_init.p, init.size() * sizeof(EF), hipMemcpyDeviceToDevice, cx.stream));
        };
        {
            EF eq_u0r0 = eq_uni(l_skip, u0, r0), eq_u0r0_rot = eq_uni(l_skip, u0, r0 * omega_skip), eq_u01 = eq_uni_at_one(l_skip, u0);
            for (size_t wi = 0; wi < W; wi++) {
                Window &wd = ws[wi];
                if (!wd.own) continue;
                size_t w0 = win[wi], w1 = win[wi + 1], hw = size_t(1) << wd.n_lift;
                EF ind = in_uni(l_skip, wd.n, u0), equ = eq_u0r0, equ_rot = eq_u0r0_rot;
                if (wd.n < 0) {
                    Fp omega = omega_skip.exp_power_of_2((unsigned)(-wd.n));
                    EF rr = r0.exp_power_of_2((unsigned)(-wd.n));
                    unsigned l = l_skip + wd.n;
                    equ = eq_uni(l, u0, rr);
                    equ_rot = eq_uni(l, u0, rr * omega);
                }
                wd.vec_a.alloc(4 * hw);
                wd.vec_b.alloc(4 * std::max<size_t>(hw / 2, 1));
                OVM_LAUNCH(stacked_tables_kernel, dim3(ceil_div(hw, 256)), dim3(256), 0, cx.stream, wd.eq_cube.p,
                                   (uint64_t)hw, ind * equ, ind * equ_rot, ind * eq_const * eq_u01, wd.vec_a.p + 2 * hw,
                                   wd.vec_a.p + 3 * hw);
                std::vector<uint64_t> off;
                std::vector<EF> ce, cr;
                for (size_t i = w0; i < w1; i++) {
                    off.push_back(((uint64_t)(loff[tvs[i].com] + tvs[i].s.col_idx) << n_stack) + (tvs[i].s.row_idx >> l_skip));
                    ce.push_back(lam[tvs[i].leq]);
                    cr.push_back(tvs[i].lrot >= 0 ? lam[(size_t)tvs[i].lrot] : EF::zero());
                }
                auto d_off = cx.upload(off);
                auto coef_pre = [](const std::vector<EF> &cs) {
                    const Fp eleven = fp(11);
                    std::vector<uint32_t> w(8 * cs.size());
                    for (size_t i = 0; i < cs.size(); i++)
                        for (int k = 0; k < 4; k++) {
                            w[8 * i + k] = cs[i].c[k].v;
                            w[8 * i + 4 + k] = k ? (cs[i].c[k] * eleven).v : 0;
                        }
                    return w;
                };
                auto d_ce = cx.upload(coef_pre(ce));
                auto d_cr = cx.upload(coef_pre(cr));
                const uint32_t chunks = stacked_round0_chunks((uint32_t)(w1 - w0), wd.n_lift);
                DevBuf<EF> scratch(chunks > 1 ? (size_t)chunks * 2 * hw : 0);
                OVM_THROW_IF(stacked_combine(q_a.p, d_off.p, (uint32_t)(w1 - w0), wd.n_lift, d_ce.p, d_cr.p, wd.vec_a.p,
                                             wd.vec_a.p + hw, scratch.p, chunks, cx.stream));
                // (no synchronisation per window: the host prepares the next window while this one runs)
                wd.cur_log_h = wd.n_lift;
                wd.in_a = true;
                if (wd.n_lift == 0) {
                    auto fin = cx.download(wd.vec_a.p, 4);
                    wd.eq_fin = fin[2];
                    wd.krot_fin = fin[3];
                    tail_init_window(wi);
                }
            }
        }
        delete ps_stf;
        wt.mark("openings.stacked.setup");
        ProfScope *ps_str = new ProfScope("stacked.mle_rounds", cx.stream);
        std::vector<std::array<EF, 2>> st_polys;
        DevBuf<EF> d_gout(2 * tvs.size() + 2);
        // per window its own partial records and 8 mailbox slots: the sums of all live windows of a round are
        // enqueued together and waited for once
        std::vector<DevBuf<EF>> round_partials(W);
        for (size_t wi = 0; wi < W; wi++)
            if (ws[wi].n_lift && ws[wi].own) round_partials[wi].alloc(std::min<size_t>(RED_BLOCKS, ((size_t(1) << (ws[wi].n_lift - 1)) + RED_THREADS - 1) / RED_THREADS) * 2 + 64);
        for (unsigned round = 1; round <= n_stack; round++) {
            std::array<EF, 2> batch{EF::zero(), EF::zero()};
            const EF *q_cur = q_in_a ? q_a.p : q_b.p;
            bool any_window_launch = false;
            for (size_t wi = 0; wi < W; wi++) {
                Window &wd = ws[wi];
                if (round > wd.n_lift || !wd.own) continue;
                const EF *vec = wd.in_a ? wd.vec_a.p : wd.vec_b.p;
                OVM_THROW_IF(stacked_round(vec, wd.cur_log_h, mb.arm(round_partials[wi].p, 8 * wi), cx.stream));
                any_window_launch = true;
            }
            // tail slices: their sums on the device, or (host loop) gather their two entries
            bool any_tail = false;
            for (size_t wi = 0; wi < W; wi++) any_tail |= round > ws[wi].n_lift && ws[wi].own;
            if (tail_on_device && any_tail) {
                OVM_THROW_IF(stacked_tail(q_cur, q_log_h, d_tail_slices.p, d_tail.p, (uint32_t)tvs.size(), l_skip, round,
                                          round > 1 ? u.back() : EF::zero(), mb.arm(tail_partials.p, 8 * W), cx.stream));
                any_window_launch = true;
            }
            std::vector<uint64_t> gidx;
            std::vector<size_t> gslice;
            for (size_t wi = 0; wi < W; wi++)
                if (!tail_on_device && round > ws[wi].n_lift && ws[wi].own)
                    for (size_t i = win[wi]; i < win[wi + 1]; i++) {
                        uint64_t row_start = (tvs[i].s.row_idx >> (l_skip + round)) << 1;
                        const uint64_t gcol = loff[tvs[i].com] + tvs[i].s.col_idx;
                        gidx.push_back((gcol << q_log_h) + row_start);
                        gidx.push_back((gcol << q_log_h) + row_start + 1);
                        gslice.push_back(i);
                    }
            std::vector<EF> gv;
            if (!gidx.empty()) {
                auto d_gi = cx.upload(gidx);
                OVM_THROW_IF(gather_ef(q_cur, d_gi.p, gidx.size(), d_gout.p, cx.stream));
                gv = cx.download(d_gout.p, gidx.size());
            }
            if (any_window_launch) mb.wait(cx.stream);
            size_t gi = 0;
            for (size_t wi = 0; wi < W; wi++) {
                Window &wd = ws[wi];
                if (!wd.own) continue;
                if (round <= wd.n_lift) {
                    auto sv = mb.read(8 * wi, 2);
                    batch[0] += sv[0];
                    batch[1] += sv[1];
                } else if (tail_on_device) {
                    continue;  // (summed below, all finished slices at once)
                } else {
                    // A finished slice contributes q(X) * eq1b(X, b) * D_i with
                    //   D_i = eq_ub_i * (lambda_eq * eq_fin + lambda_rot * krot_fin)   (constant in X),
                    //   q(1) = t1, q(2) = 2 t1 - t0,  eq1b(1, b) = b,  eq1b(2, b) = b ? 2 : -1
                    // -- two extension-field multiplies per slice and round (this loop runs over every
                    // stacked column of a many-AIR proof, with the device idle).
                    for (size_t i = win[wi]; i < win[wi + 1]; i++, gi++) {
                        const EF t0 = gv[2 * gi], t1 = gv[2 * gi + 1];
                        const size_t b = (tvs[i].s.row_idx >> (l_skip + round - 1)) & 1;
                        if (!tail_ready[i]) {
                            tail_d[i] = lam[tvs[i].leq] * wd.eq_fin;
                            if (tvs[i].lrot >= 0) tail_d[i] += lam[(size_t)tvs[i].lrot] * wd.krot_fin;
                            tail_ready[i] = 1;
                        }
                        const EF v2 = (t1 + t1 - t0) * tail_d[i];
                        if (b) {
                            batch[0] += t1 * tail_d[i];
                            batch[1] += v2 + v2;
                        } else {
                            batch[1] -= v2;
                        }
                    }
                }
            }
            if (tail_on_device && any_tail) {
                auto sv = mb.read(8 * W, 2);
                batch[0] += sv[0];
                batch[1] += sv[1];
            }
            if (!sh.single()) {  // every rank has summed its own windows and finished slices
                std::vector<EF> bv{batch[0], batch[1]};
                share_sum(bv);
                batch = {bv[0], bv[1]};
            }
            for (auto &e : batch) ts.observe_ext(e);
            st_polys.push_back(batch);
            EF ur = ts.sample_ext();
            u.push_back(ur);
            {
                EF *qo = q_in_a ? q_b.p : q_a.p;
                if (lwidth) OVM_THROW_IF(fold_mle(q_cur, qo, (uint32_t)lwidth, q_log_h, ur, cx.stream));
                q_in_a = !q_in_a;
                q_log_h--;
            }
            {
                // the four vectors of every live window fold in one launch (big windows on their own)
                std::vector<FoldJob> fj;
                for (size_t wi = 0; wi < W; wi++) {
                    Window &wd = ws[wi];
                    if (wd.cur_log_h == 0 || !wd.own) continue;
                    const EF *in = wd.in_a ? wd.vec_a.p : wd.vec_b.p;
                    EF *o = wd.in_a ? wd.vec_b.p : wd.vec_a.p;
                    if (wd.cur_log_h <= 11)
                        fj.push_back(FoldJob{in, o, 4, wd.cur_log_h, fj.empty() ? 0 : fj.back().y0 + fj.back().n_cols, 0});
                    else
                        OVM_THROW_IF(fold_mle(in, o, 4, wd.cur_log_h, ur, cx.stream));
                }
                if (!fj.empty() && fj.size() <= FoldJobsArg::MAX) {
                    OVM_THROW_IF(fold_mle_batched(nullptr, fj.data(), (uint32_t)fj.size(), ur, cx.stream));
                } else if (!fj.empty()) {
                    auto d_fj = cx.upload(fj);
                    OVM_THROW_IF(fold_mle_batched(d_fj.p, fj.data(), (uint32_t)fj.size(), ur, cx.stream));
                }
                for (size_t wi = 0; wi < W; wi++) {
                    Window &wd = ws[wi];
                    if (wd.cur_log_h == 0 || !wd.own) continue;
                    EF *o = wd.in_a ? wd.vec_b.p : wd.vec_a.p;
                    wd.in_a = !wd.in_a;
                    wd.cur_log_h--;
                    if (wd.cur_log_h == 0) {
                        auto fin = cx.download(o, 4);
                        wd.eq_fin = fin[2];
                        wd.krot_fin = fin[3];
                        tail_init_window(wi);
                    }
                }
            }
            for (size_t i = 0; i < tvs.size() && !tail_on_device; i++) {
                unsigned n_lift = tvs[i].s.log_height > l_skip ? tvs[i].s.log_height - l_skip : 0;
                if (round > n_lift && tail_ready[i]) tail_d[i] = tail_d[i] * eq1b(ur, (tvs[i].s.row_idx >> (l_skip + round - 1)) & 1);
            }
        }
        delete ps_str;
        wt.mark("openings.stacked.rounds");
        std::vector<EF> stacking_openings(width, EF::zero());
        {
            std::vector<EF> local = cx.download(q_in_a ? q_a.p : q_b.p, lwidth);
            std::vector<std::vector<EF>> per_commit(commits.size());
            for (size_t com = 0; com < commits.size(); com++)
                if (own(com)) per_commit[com].assign(local.begin() + loff[com], local.begin() + loff[com] + commits[com].layout.width);
            share_vectors(per_commit, [&](size_t com) { return commits[com].owner; });
            for (size_t com = 0; com < commits.size(); com++) {
                if (per_commit[com].size() != commits[com].layout.width) throw OvmFail(1, "shard exchange: stacking openings of the wrong length");
                std::copy(per_commit[com].begin(), per_commit[com].end(), stacking_openings.begin() + coff[com]);
            }
        }
        for (auto &c : stacking_openings) ts.observe_ext(c);
        q_a.release();
        q_b.release();
        for (auto &wd : ws) {
            wd.vec_a.release();
            wd.vec_b.release();
            wd.eq_cube.release();
        }
        // stacking proof bytes
        w.e_slice(s0);
        w.usize(st_polys.size());
        for (auto &a : st_polys)
            for (auto &x : a) w.e(x);
        w.usize(commits.size());
        for (size_t com = 0; com < commits.size(); com++)
            w.e_slice(std::vector<EF>(stacking_openings.begin() + coff[com], stacking_openings.begin() + coff[com + 1]));
        section_offsets.push_back(w.buf.size());
        return u;
    }

    // WHIR over the mu-batch of every commitment's columns. Sharded: every rank batches ITS columns, the vectors
    // are all-gathered and summed, rank 0 runs the protocol and collects the rows the first round opens.
    void prove_whir(const std::vector<CommitRef> &commits, const std::vector<size_t> &coff, const std::vector<EF> &u) {
        const unsigned l_skip = P().l_skip, n_stack = P().n_stack;
        const bool root_rank = sh.rank == 0;
        auto own = [&](size_t com) { return commits[com].owner == sh.rank; };
        const size_t width = coff.back();
        ByteWriter &w = out;
        std::vector<EF> u_cube;
        {
            EF cur = u[0];
            for (unsigned i = 0; i < l_skip; i++) {
                u_cube.push_back(cur);
                cur = cur.square();
            }
            u_cube.insert(u_cube.end(), u.begin() + 1, u.end());
        }
        const unsigned k_whir = P().k_whir, log_blowup = P().log_blowup;
        const size_t n_rounds = P().whir_num_queries.size();
        Fp mu_pow_witness = ts.grind(P().mu_pow_bits);
        wt.mark("openings.whir.mu_pow");
        EF mu = ts.sample_ext();
        unsigned m = l_skip + n_stack;
        // (sharded with an all-to-all: the vector is exchanged in `world` slices of ceil(2^m / world) elements)
        const size_t mu_slice = sh.has_alltoall() ? ((size_t(1) << m) + sh.world - 1) / sh.world : 0;
        const size_t mu_padded = std::max(size_t(1) << m, mu_slice * sh.world);
        DevBuf<EF> f_a(mu_padded), f_b(size_t(1) << (m - 1)), w_a, w_b;
        {
            ProfScope ps_mu("whir.mu_batch", cx.stream);
            std::vector<EF> mu_pows = powers_of(mu, width);
            auto d_mu = cx.upload(mu_pows);
            bool first = true;
            for (size_t com = 0; com < commits.size(); com++) {
                if (!own(com)) continue;
                OVM_THROW_IF(whir_mu_batch(commits[com].pcs->msg.p, (uint32_t)commits[com].layout.width, m, d_mu.p,
                                           (uint32_t)coff[com], f_a.p, first ? 0 : 1, cx.stream));
                first = false;
            }
            if (first) OVM_THROW_IF(hipMemsetAsync(f_a.p, 0, sizeof(EF) << m, cx.stream));  // (a rank without commitments)
        }
        if (wt.on) cx.sync();
        wt.mark("openings.whir.mu_batch");
        if (!sh.single()) {
            // THE bulk exchange of a sharded proof (north_star: "RCCL allgather only at the FRI batch-opening
            // step"; prover/whir.rs:95-132 batches all commitments into one vector): every rank holds the
            // mu-weighted sum of ITS columns and every rank needs the sum over the ranks (modular addition is not
            // an RCCL reduction). EVERY rank then runs the WHIR rounds on the sum (the folds are cheap; the
            // transcript stays replicated) so that the rounds' Merkle trees -- half of WHIR's time -- can be built
            // by all ranks together (build_tree); rank 0 alone assembles the proof.
            OVM_PROF("whir.mu_allgather", cx.stream);
            if (sh.has_alltoall()) {
                // all-to-all (rank r receives slice r of every rank's vector) + add + all-gather of the summed
                // slices: 2 (N-1)/N vectors through every GPU's links instead of N-1, and xGMI is a full mesh of
                // point-to-point links, which is the pattern an all-to-all uses best
                if (mu_padded > (size_t(1) << m))
                    OVM_THROW_IF(hipMemsetAsync(f_a.p + (size_t(1) << m), 0, (mu_padded - (size_t(1) << m)) * sizeof(EF), cx.stream));
                DevBuf<EF> recv(mu_padded), red(mu_slice);
                sh.alltoall_device(f_a.p, recv.p, mu_slice * sizeof(EF), (ovm_stream_t)cx.stream);
                OVM_THROW_IF(ef_sum_blocks(recv.p, sh.world, mu_slice, red.p, cx.stream));
                sh.allgather_device(red.p, f_a.p, mu_slice * sizeof(EF), (ovm_stream_t)cx.stream);
            } else {
                DevBuf<EF> gathered((size_t)sh.world << m);
                sh.allgather_device(f_a.p, gathered.p, sizeof(EF) << m, (ovm_stream_t)cx.stream);
                OVM_THROW_IF(ef_sum_blocks(gathered.p, sh.world, uint64_t(1) << m, f_a.p, cx.stream));
            }
        }
        if (wt.on) cx.sync();
        wt.mark("openings.whir.mu_exchange");
        w_a.alloc(size_t(1) << m);
        w_b.alloc(size_t(1) << (m - 1));
        {
            OVM_PROF("whir.mu_batch", cx.stream);
            OVM_THROW_IF(zeta_transform(f_a.p, m, +1, cx.stream));
            auto d_u = cx.upload(u_cube);
            OVM_THROW_IF(mobius_eq_table(d_u.p, m, w_a.p, cx.stream));
            cx.sync();
        }
        bool fw_in_a = true;
        std::vector<std::array<EF, 2>> whir_polys;
        std::vector<p2::Digest> codeword_commits;
        std::vector<EF> ood_values, final_poly;
        std::vector<Fp> folding_pows, query_pows;
        // per commitment, per query: 2^k rows x that commitment's width / the Merkle path
        std::vector<std::vector<std::vector<std::vector<Fp>>>> init_rows(commits.size());
        std::vector<std::vector<std::vector<p2::Digest>>> init_proofs(commits.size());
        std::vector<std::vector<std::vector<EF>>> cw_values;        // per round, per query: 2^k
        std::vector<std::vector<std::vector<p2::Digest>>> cw_proofs;
        unsigned log_rs = m + log_blowup;
        PcsData g_prev;  // previous round's codeword tree (EF, width 1)
        DevBuf<EF> d_red((size_t)RED_BLOCKS * 2 + 8);
        for (size_t wr = 0; wr < n_rounds; wr++) {
            const bool is_last = wr + 1 == n_rounds;
            ProfScope *ps_ws = new ProfScope("whir.sumcheck_fold", cx.stream);
            EF al = EF::zero();
            for (unsigned rd = 0; rd < k_whir; rd++) {
                const EF *f = fw_in_a ? f_a.p : f_b.p, *wv = fw_in_a ? w_a.p : w_b.p;
                uint64_t half = 1ull << (m - rd - 1);
                if (rd == 0) {
                    OVM_THROW_IF(whir_round(f, wv, half, mb.arm(d_red.p), cx.stream));
                } else {
                    // fold round rd-1 with its alpha and get this round's sums in the same pass
                    OVM_THROW_IF(whir_fold_round(f, wv, fw_in_a ? f_b.p : f_a.p, fw_in_a ? w_b.p : w_a.p, half, al,
                                                 mb.arm(d_red.p), cx.stream));
                    fw_in_a = !fw_in_a;
                }
                auto sv = mb.take(2, cx.stream);
                ts.observe_ext(sv[0]);
                ts.observe_ext(sv[1]);
                whir_polys.push_back({sv[0], sv[1]});
                folding_pows.push_back(ts.grind(P().folding_pow_bits));
                al = ts.sample_ext();
            }
            {
                // the last fold of this WHIR round
                const EF *f = fw_in_a ? f_a.p : f_b.p, *wv = fw_in_a ? w_a.p : w_b.p;
                OVM_THROW_IF(whir_fold(f, wv, fw_in_a ? f_b.p : f_a.p, fw_in_a ? w_b.p : w_a.p, 1ull << (m - k_whir), al, cx.stream));
                fw_in_a = !fw_in_a;
            }
            delete ps_ws;
            ProfScope *ps_wc = new ProfScope("whir.rs_commit", cx.stream);
            const unsigned m_new = m - k_whir;
            EF *f_cur = fw_in_a ? f_a.p : f_b.p, *w_cur = fw_in_a ? w_a.p : w_b.p;
            // g = f(alpha, .) in coefficient form
            DevBuf<EF> g_coeffs(size_t(1) << m_new);
            OVM_THROW_IF(hipMemcpyAsync(g_coeffs.p, f_cur, sizeof(EF) << m_new, hipMemcpyDeviceToDevice, cx.stream));
            OVM_THROW_IF(zeta_transform(g_coeffs.p, m_new, -1, cx.stream));
            PcsData g_tree;
            EF z0 = EF::zero();
            if (!is_last) {
                g_tree.log_rows = log_rs - 1;
                g_tree.k = k_whir;
                DevBuf<EF> g_rs(size_t(1) << g_tree.log_rows);
                OVM_THROW_IF(ovm_hip_ntt_dif_ext((const uint32_t *)g_coeffs.p, (uint32_t *)g_rs.p, 1, g_tree.log_rows, m_new, cx.stream));
                build_tree(g_tree, (const uint32_t *)g_rs.p, 1, true, /*distributed=*/true);
                g_tree.codeword = DevBuf<Fp>();
                g_tree_codeword = std::move(g_rs);
                ts.observe_digest(g_tree.root);
                codeword_commits.push_back(g_tree.root);
                z0 = ts.sample_ext();
                OVM_THROW_IF(horner_eval(g_coeffs.p, 1ull << m_new, z0, mb.arm(d_red.p), cx.stream));
                EF opened = mb.take(1, cx.stream)[0];
                ts.observe_ext(opened);
                ood_values.push_back(opened);
            } else {
                final_poly = cx.download(g_coeffs.p, size_t(1) << m_new);
                for (auto &c : final_poly) ts.observe_ext(c);
            }
            delete ps_wc;
            const unsigned lq = log_rs - k_whir;
            Fp omega = bb::two_adic_generator(lq);
            size_t nq = P().whir_num_queries[wr];
            query_pows.push_back(ts.grind(P().query_phase_pow_bits));
            std::vector<uint64_t> qidx(nq);
            for (auto &q : qidx) q = ts.sample_bits(lq);
            // opened rows + Merkle paths. Rows come from their commitment's owner (round 0) or from this rank's own copy
            // of the previous round's codeword; path digests from whichever ranks' layers hold them (path_refs).
            {
                OVM_PROF("whir.queries", cx.stream);
                // per tree that is opened in this round: [rank] -> the digests that rank's layers hold
                auto exchange = [&](const Packer &mine) -> std::vector<std::vector<uint8_t>> {
                    if (sh.single()) return {};
                    return sh.allgatherv(root_rank ? std::vector<uint8_t>() : mine.buf);
                };
                auto assemble = [&](const PcsData &pcs, const std::vector<PathRef> &refs, std::vector<std::vector<p2::Digest>> &served) {
                    std::vector<p2::Digest> digs;
                    std::vector<size_t> at(sh.world, 0);
                    for (auto &rf : refs) {
                        if (rf.rank < 0) {
                            digs.push_back(pcs.top.at(rf.idx));
                        } else {
                            auto &q = served.at(rf.rank);
                            if (at[rf.rank] >= q.size()) throw OvmFail(1, "shard exchange: Merkle path digests missing");
                            digs.push_back(q[at[rf.rank]++]);
                        }
                    }
                    return digs;
                };
                if (wr == 0) {
                    // this rank's part: the rows of its commitments, the digests its layers hold of every tree
                    Packer mine;
                    std::vector<std::vector<Fp>> vals_of(commits.size());
                    std::vector<std::vector<std::vector<p2::Digest>>> served(commits.size(), std::vector<std::vector<p2::Digest>>(sh.world));
                    std::vector<std::vector<PathRef>> refs_of(commits.size());
                    for (size_t com = 0; com < commits.size(); com++) {
                        const PcsData *pcs = commits[com].pcs;
                        if (!pcs || !(own(com) || pcs->dist_n > 1)) continue;
                        if (own(com)) open_rows(*pcs, qidx, lq, vals_of[com]);
                        refs_of[com] = path_refs(*pcs, (int)commits[com].owner, qidx, lq);
                        served[com][sh.rank] = serve_paths(*pcs, refs_of[com]);
                        mine.put<uint64_t>(com);
                        mine.put_vec(vals_of[com]);
                        mine.put_vec(served[com][sh.rank]);
                    }
                    const auto remote = exchange(mine);
                    if (root_rank) {
                        for (uint32_t rk = 1; rk < sh.world; rk++) {
                            Unpacker un(remote[rk]);
                            while (!un.done()) {
                                const uint64_t com = un.get<uint64_t>();
                                if (com >= commits.size()) throw OvmFail(1, "shard exchange: unexpected opened rows");
                                std::vector<Fp> v = un.get_vec<Fp>();
                                if (commits[com].owner == rk) vals_of[com] = std::move(v);
                                served[com][rk] = un.get_vec<p2::Digest>();
                            }
                        }
                        for (size_t com = 0; com < commits.size(); com++) {
                            const PcsData *pcs = commits[com].pcs;
                            const size_t cw = commits[com].layout.width;
                            // a tree held whole by its (remote) owner: its paths arrive complete
                            const std::vector<p2::Digest> digs = (!own(com) && !(pcs && pcs->dist_n > 1))
                                                                     ? std::move(served[com][commits[com].owner])
                                                                     : assemble(*pcs, refs_of[com], served[com]);
                            const auto &vals = vals_of[com];
                            if (vals.size() != (nq << k_whir) * cw || digs.size() != nq * lq) throw OvmFail(1, "opened rows of the wrong size");
                            for (size_t q = 0; q < nq; q++) {
                                std::vector<std::vector<Fp>> qr;
                                for (size_t t = 0; t < (size_t(1) << k_whir); t++) {
                                    size_t ro = (q << k_whir) + t;
                                    qr.emplace_back(vals.begin() + ro * cw, vals.begin() + (ro + 1) * cw);
                                }
                                init_rows[com].push_back(std::move(qr));
                                init_proofs[com].emplace_back(digs.begin() + q * lq, digs.begin() + (q + 1) * lq);
                            }
                        }
                    }
                } else {
                    const std::vector<PathRef> refs = path_refs(g_prev, 0, qidx, lq);
                    std::vector<std::vector<p2::Digest>> served(sh.world);
                    served[sh.rank] = serve_paths(g_prev, refs);
                    Packer mine;
                    mine.put_vec(served[sh.rank]);
                    const auto remote = g_prev.dist_n > 1 ? exchange(mine) : std::vector<std::vector<uint8_t>>();
                    if (root_rank) {
                        for (uint32_t rk = 1; rk < sh.world && !remote.empty(); rk++) {
                            Unpacker un(remote[rk]);
                            served[rk] = un.get_vec<p2::Digest>();
                        }
                        const std::vector<p2::Digest> digs = assemble(g_prev, refs, served);
                        std::vector<uint64_t> rows;
                        for (uint64_t i : qidx) {
                            uint64_t grp = bitrev64(i, lq);
                            for (uint64_t t = 0; t < (1ull << k_whir); t++) rows.push_back((grp << k_whir) + bitrev64(t, k_whir));
                        }
                        auto d_rows = cx.upload(rows);
                        DevBuf<EF> d_vals(std::max<size_t>(rows.size(), 1));
                        OVM_THROW_IF(gather_ef(prev_codeword.p, d_rows.p, rows.size(), d_vals.p, cx.stream));
                        auto vals = cx.download(d_vals.p, rows.size());
                        std::vector<std::vector<EF>> rv;
                        std::vector<std::vector<p2::Digest>> rp;
                        for (size_t q = 0; q < nq; q++) {
                            rv.emplace_back(vals.begin() + (q << k_whir), vals.begin() + ((q + 1) << k_whir));
                            rp.emplace_back(digs.begin() + q * lq, digs.begin() + (q + 1) * lq);
                        }
                        cw_values.push_back(std::move(rv));
                        cw_proofs.push_back(std::move(rp));
                    }
                }
            }
            g_prev = std::move(g_tree);
            prev_codeword = std::move(g_tree_codeword);
            EF gamma = ts.sample_ext();
            if (!is_last) {
                // w += gamma * eq(., pow(z0)) + sum_i gamma^(i+2) * eq(., pow(z_i))
                OVM_PROF("whir.w_accumulate", cx.stream);
                const size_t n_q = nq + 1;
                std::vector<EF> zp(n_q * m_new), scale(n_q);
                {
                    EF cur = z0;
                    for (unsigned j = 0; j < m_new; j++) {
                        zp[j] = cur;
                        cur = cur.square();
                    }
                    scale[0] = gamma;
                    EF gp = gamma * gamma;
                    for (size_t q = 0; q < nq; q++) {
                        Fp zc = omega.pow(qidx[q]);
                        for (unsigned j = 0; j < m_new; j++) {
                            zp[(q + 1) * m_new + j] = ef(zc);
                            zc = zc.square();
                        }
                        scale[q + 1] = gp;
                        gp = gp * gamma;
                    }
                }
                unsigned log_lo = m_new / 2, log_hi = m_new - log_lo;
                auto d_zp = cx.upload(zp);
                auto d_sc = cx.upload(scale);
                DevBuf<EF> tlo(n_q << log_lo), thi(n_q << log_hi);
                OVM_LAUNCH(whir_eq_tables_kernel, dim3(ceil_div((uint64_t)n_q << log_lo, 256)), dim3(256), 0, cx.stream,
                                   d_zp.p, m_new, 0u, log_lo, d_sc.p, (uint32_t)n_q, tlo.p);
                OVM_LAUNCH(whir_eq_tables_kernel, dim3(ceil_div((uint64_t)n_q << log_hi, 256)), dim3(256), 0, cx.stream,
                                   d_zp.p, m_new, log_lo, log_hi, (const EF *)nullptr, (uint32_t)n_q, thi.p);
                OVM_THROW_IF(whir_w_accumulate(w_cur, m_new, tlo.p, thi.p, log_lo, (uint32_t)n_q, cx.stream));
                cx.sync();
            }
            m = m_new;
            log_rs -= 1;
        }
        if (!root_rank) {  // (the proof is assembled on rank 0)
            section_offsets.push_back(w.buf.size());
            return;
        }
        // whir proof bytes
        w.f(mu_pow_witness);
        w.usize(whir_polys.size());
        for (auto &a : whir_polys)
            for (auto &x : a) w.e(x);
        w.usize(codeword_commits.size());
        for (auto &d : codeword_commits) w.digest(d);
        for (auto &x : ood_values) w.e(x);
        for (auto &x : folding_pows) w.f(x);
        for (auto &x : query_pows) w.f(x);
        w.usize(commits.size());
        w.usize(init_rows[0].size());
        if (!init_rows[0].empty()) {
            w.usize(init_proofs[0][0].size());
            for (auto &c : commits) w.usize(c.layout.width);
            for (auto &cr : init_rows)
                for (auto &qr : cr)
                    for (auto &row : qr)
                        for (auto &x : row) w.f(x);
            for (auto &cp : init_proofs)
                for (auto &pr : cp)
                    for (auto &d : pr) w.digest(d);
        }
        for (auto &round : cw_values) {
            w.usize(round.size());
            for (auto &q : round)
                for (auto &x : q) w.e(x);
        }
        size_t first_depth = 0;
        if (n_rounds > 1 && !init_rows[0].empty()) first_depth = cw_proofs[0][0].size();
        w.usize(first_depth);
        for (auto &round : cw_proofs)
            for (auto &pr : round)
                for (auto &d : pr) w.digest(d);
        w.e_slice(final_poly);
        section_offsets.push_back(w.buf.size());
    }
    DevBuf<EF> g_tree_codeword, prev_codeword;

    // -----------------------------------------------------------------------------------------
    // Coordinator::prove
    // -----------------------------------------------------------------------------------------
    // ProvingContext::into_sorted (prover/types.rs:137-148): height descending, AIR index ascending
    void set_traces(std::vector<TraceIn> in, bool need_public_values = true) {
        traces = std::move(in);
        std::stable_sort(traces.begin(), traces.end(), [](const TraceIn &a, const TraceIn &b) {
            if (a.height != b.height) return a.height > b.height;
            return a.air_idx < b.air_idx;
        });
        owner.clear();
        for (auto &t : traces) {
            if (t.air_idx >= pk.airs.size()) throw OvmFail(1, "air index out of range");
            if (t.owner >= sh.world) throw OvmFail(1, "trace owner out of range");
            if (!owner.empty() && t.owner < owner.back())
                throw OvmFail(1, "sharded proving: owners must be non-decreasing over the traces sorted by (height desc, air index)");
            owner.push_back(t.owner);
            if (need_public_values && t.owner == sh.rank && t.public_values.size() != pk.airs[t.air_idx].num_public_values)
                throw OvmFail(1, "public values length");
            if (t.owner == sh.rank) {
                t.cached_roots.clear();
                for (auto *c : t.cached) t.cached_roots.push_back(c->root);
            }
        }
    }

    // Sharded: the common-main commitment stacks the common mains of ALL traces. Every rank contributes the
    // common mains of ITS traces (concatenated in sorted order) to one device all-gather and commits from the
    // gathered buffer, with the Merkle tree built by all ranks together. (The per-AIR bulk -- the cached mains --
    // never moves: SURVEY §8(e) "committing each AIR's trace as its own commitment".)
    DevBuf<Fp> common_gathered;
    void commit_common_sharded(PcsData &pcs, p2::Digest &root) {
        std::vector<uint64_t> words_of_rank(sh.world, 0), off_in_rank(traces.size(), 0);
        for (size_t t = 0; t < traces.size(); t++) {
            off_in_rank[t] = words_of_rank[owner[t]];
            words_of_rank[owner[t]] += (uint64_t)air(t).common_main_width * traces[t].height;
        }
        uint64_t max_words = 0;
        for (uint64_t wds : words_of_rank) max_words = std::max(max_words, wds);
        if (max_words) {
            DevBuf<Fp> send(max_words);
            for (size_t t = 0; t < traces.size(); t++) {
                const uint64_t n = (uint64_t)air(t).common_main_width * traces[t].height;
                if (mine(t) && n)
                    OVM_THROW_IF(hipMemcpyAsync(send.p + off_in_rank[t], traces[t].d_trace, n * sizeof(Fp), hipMemcpyDeviceToDevice, cx.stream));
            }
            common_gathered.alloc(max_words * sh.world);
            sh.allgather_device(send.p, common_gathered.p, max_words * sizeof(Fp), (ovm_stream_t)cx.stream);
            cx.sync();  // (`send` goes back to the cache: the exchange may have run on another stream)
        }
        // Every rank now holds every common main: each stacks and encodes them (a column or two) and hashes ITS share of
        // the codeword's query groups; the shares' roots are exchanged (build_tree). Rank 0 keeps owning the
        // commitment's columns in the stacked reduction and the mu-batch.
        std::vector<MatIn> mats;
        for (size_t t = 0; t < traces.size(); t++)
            mats.push_back({common_gathered.p + owner[t] * max_words + off_in_rank[t], air(t).common_main_width, traces[t].height});
        pcs = commit_mats(mats, true);
        root = pcs.root;
    }
    // Sharded: public values and cached-main commitments of every trace, from its owner.
    void exchange_trace_meta() {
        Packer pk_;
        for (size_t t = 0; t < traces.size(); t++)
            if (mine(t)) {
                pk_.put<uint64_t>(t);
                pk_.put_vec(traces[t].public_values);
                pk_.put_vec(traces[t].cached_roots);
            }
        const auto all = sh.allgatherv(pk_.buf);
        for (uint32_t r = 0; r < sh.world; r++) {
            if (r == sh.rank) continue;
            Unpacker u(all[r]);
            while (!u.done()) {
                const uint64_t t = u.get<uint64_t>();
                if (t >= traces.size() || owner[t] != r) throw OvmFail(1, "shard exchange: unexpected trace record");
                traces[t].public_values = u.get_vec<Fp>();
                traces[t].cached_roots = u.get_vec<p2::Digest>();
            }
        }
        for (size_t t = 0; t < traces.size(); t++)
            if (traces[t].public_values.size() != air(t).num_public_values || traces[t].cached_roots.size() != air(t).cached_widths.size())
                throw OvmFail(1, "shard exchange: public values / cached commitments of a trace are missing");
    }

    void prove(std::vector<TraceIn> in) {
        set_traces(std::move(in));
        prove_loaded();
    }
    // (after set_traces: the sharded entry point lets the ranks agree on their validation results in between)
    void prove_loaded() {
        ts.observe_digest(pk.vk_pre_hash);
        programs_future = std::async(std::launch::async, [this, tn = tuning()] {
            Tuning::current() = tn;  // (thread-local: the helper works under the caller's settings)
            return compile_all_programs();
        });
        wt.mark("setup");
        PcsData pcs;
        p2::Digest common_root;
        if (sh.single()) {
            pcs = commit();
            common_root = pcs.root;
        } else {
            commit_common_sharded(pcs, common_root);
            exchange_trace_meta();
            if (tuning().test_fail_rank == (int)sh.rank)  // (tests: a rank that dies in the middle of a joint proof)
                throw OvmFail((int)hipErrorOutOfMemory, "test hook OVM_TEST_FAIL_RANK: simulated failure after the commitment");
        }
        wt.mark("commit");
        const size_t num_airs = pk.airs.size();
        std::vector<int> present(num_airs, -1);
        for (size_t t = 0; t < traces.size(); t++) present[traces[t].air_idx] = (int)t;
        ts.observe_digest(common_root);
        for (size_t a = 0; a < num_airs; a++) {
            if (!pk.airs[a].is_required) ts.observe(present[a] >= 0 ? Fp::one() : Fp::zero());
            if (present[a] >= 0) {
                const size_t t = (size_t)present[a];
                // prover/mod.rs:156-170: a preprocessed commitment stands in for the height
                if (pk.airs[a].preprocessed_width) {
                    if (!pre_of(t)) throw OvmFail(1, "preprocessed commitment missing");
                    ts.observe_digest(pre_of(t)->root);
                } else {
                    ts.observe(fp(log2_strict(traces[t].height)));
                }
                for (auto &c : traces[t].cached_roots) ts.observe_digest(c);
                for (auto &pv : traces[t].public_values) ts.observe(pv);
            }
        }
        // proof header
        out.u32(3);  // CODEC_VERSION
        out.digest(common_root);
        out.usize(num_airs);
        for (size_t a = 0; a < num_airs; a += 8) {
            uint8_t b = 0;
            for (size_t j = 0; j < 8 && a + j < num_airs; j++)
                if (present[a + j] >= 0) b |= (uint8_t)(1u << j);
            out.u8(b);
        }
        for (size_t a = 0; a < num_airs; a++)
            if (present[a] >= 0) {
                out.usize(log2_strict(traces[present[a]].height));
                out.usize(traces[present[a]].cached_roots.size());
                for (auto &c : traces[present[a]].cached_roots) out.digest(c);
            }
        out.usize(num_airs);
        for (size_t a = 0; a < num_airs; a++) {
            if (present[a] >= 0) {
                out.usize(traces[present[a]].public_values.size());
                for (auto &pv : traces[present[a]].public_values) out.f(pv);
            } else {
                out.usize(0);
            }
        }
        section_offsets.push_back(out.buf.size());
        std::vector<EF> r = prove_rap_constraints();
        wt.mark("rap_constraints.rest");
        prove_openings(&pcs, common_root, r);
        wt.mark("openings.whir");
    }
};


// Byte offsets of the ends of the five proof sections (header, GKR, batch constraints, stacking, WHIR)
// for traces of the given shapes: the encoding (proof.rs:226-421) has no data-dependent lengths, so the
// size of a proof is a function of the key and the trace heights alone. `ovm_hip_proof_size` exposes it
// so that callers size their buffers once (reference: the buffer-size queries of
// crates/cuda-backend/src/cuda/logup_zerocheck.rs:152,344-352).
struct ShapeIn {
    uint32_t air_idx;
    size_t height;
};
inline std::array<uint64_t, 5> proof_section_ends(const ProvingKey &pk, std::vector<ShapeIn> tr) {
    const Params &P = pk.params;
    std::stable_sort(tr.begin(), tr.end(), [](const ShapeIn &a, const ShapeIn &b) {
        if (a.height != b.height) return a.height > b.height;
        return a.air_idx < b.air_idx;
    });
    const unsigned l_skip = P.l_skip, n_stack = P.n_stack, k = P.k_whir;
    const uint64_t N = uint64_t(1) << l_skip;
    const size_t num_airs = pk.airs.size(), T = tr.size();
    std::vector<char> present(num_airs, 0);
    for (auto &t : tr) {
        if (t.air_idx >= num_airs) throw OvmFail(1, "air index out of range");
        if (present[t.air_idx]) throw OvmFail(1, "duplicate air index");
        if (t.height == 0 || (t.height & (t.height - 1))) throw OvmFail(1, "trace height");
        present[t.air_idx] = 1;
    }
    std::array<uint64_t, 5> ends{};
    uint64_t sz = 4 + 32 + 4 + (num_airs + 7) / 8;
    for (auto &t : tr) sz += 8 + 32 * pk.airs[t.air_idx].cached_widths.size();
    sz += 4;
    for (size_t a = 0; a < num_airs; a++) sz += 4 + (present[a] ? 4 * (uint64_t)pk.airs[a].num_public_values : 0);
    ends[0] = sz;
    // GKR
    uint64_t total_interactions = 0;
    unsigned n_max = 0;
    for (auto &t : tr) {
        const unsigned lh = log2_strict(t.height), lifted = std::max(lh, l_skip);
        total_interactions += (uint64_t)pk.airs[t.air_idx].interactions.size() << lifted;
        n_max = std::max(n_max, lifted - l_skip);
    }
    sz += 4 + 16 + 4;
    if (total_interactions) {
        const unsigned TR = 64 - (unsigned)__builtin_clzll(total_interactions);  // l_skip + n_logup
        sz += 64ull * TR;
        for (unsigned round = 1; round < TR; round++) sz += 48ull * round;
    }
    ends[1] = sz;
    // batch constraints
    const uint64_t cdeg = P.max_constraint_degree, s_deg = cdeg + 1, s0_deg = s_deg * (N - 1);
    sz += 4 + 16 * T + 16 * T + 4 + 16 * (s0_deg + 1) + 4;
    if (n_max) sz += 4 + 16ull * n_max * s_deg;
    for (auto &t : tr) {
        const AirKey &a = pk.airs[t.air_idx];
        const uint64_t rot = a.need_rot ? 2 : 1;
        sz += 4 + 4 + 16 * rot * a.common_main_width;
        if (a.preprocessed_width) sz += 4 + 16 * rot * a.preprocessed_width;
        for (size_t w : a.cached_widths) sz += 4 + 16 * rot * w;
    }
    ends[2] = sz;
    // stacking: commitments = common main, then per trace preprocessed + cached mains
    std::vector<uint64_t> widths;
    {
        std::vector<std::pair<size_t, unsigned>> meta;
        for (auto &t : tr) meta.push_back({pk.airs[t.air_idx].common_main_width, log2_strict(t.height)});
        widths.push_back(Layout(l_skip, l_skip + n_stack, meta).width);
        for (auto &t : tr) {
            const AirKey &a = pk.airs[t.air_idx];
            const unsigned lh = log2_strict(t.height);
            if (a.preprocessed_width) widths.push_back(Layout(l_skip, l_skip + n_stack, {{a.preprocessed_width, lh}}).width);
            for (size_t w : a.cached_widths) widths.push_back(Layout(l_skip, l_skip + n_stack, {{w, lh}}).width);
        }
    }
    sz += 4 + 16 * (2 * (N - 1) + 1) + 4 + 32ull * n_stack + 4;
    for (uint64_t w : widths) sz += 4 + 16 * w;
    ends[3] = sz;
    // WHIR
    const size_t n_rounds = P.whir_num_queries.size();
    unsigned m = l_skip + n_stack, log_rs = m + P.log_blowup;
    sz += 4 + 4 + 32ull * n_rounds * k + 4;
    if (n_rounds) sz += 48ull * (n_rounds - 1);  // codeword commitments + out-of-domain values
    sz += 4ull * n_rounds * k + 4ull * n_rounds + 4 + 4;
    const uint64_t nq0 = n_rounds ? P.whir_num_queries[0] : 0;
    if (nq0) {
        sz += 4 + 4 * widths.size();
        for (uint64_t w : widths) sz += ((nq0 << k) * w) * 4 + nq0 * (log_rs - k) * 32;
    }
    for (size_t wr = 1; wr < n_rounds; wr++) sz += 4 + (P.whir_num_queries[wr] << k) * 16 + P.whir_num_queries[wr] * (log_rs - wr - k) * 32;
    sz += 4;
    if (m < n_rounds * k) throw OvmFail(1, "WHIR rounds exceed the stacked height");
    sz += 4 + (16ull << (m - n_rounds * k));
    ends[4] = sz;
    return ends;
}

}  // namespace ovm

using namespace ovm;

static thread_local std::string g_last_error;

bool ovm::Tuning::live() {
    static const bool on = [] {
        const char *e = getenv("OVM_TUNING_LIVE");
        return e && *e && *e != '0';
    }();
    return on;
}
Tuning ovm::Tuning::from_env() {
    Tuning t;
    auto num = [](const char *name, int dflt) {
        const char *e = getenv(name);
        return e && *e ? atoi(e) : dflt;
    };
    auto off = [](const char *name) {  // "=0" switches a default-on feature off
        const char *e = getenv(name);
        return e && e[0] == '0';
    };
    t.r0_rows = num("OVM_R0_ROWS", 0);
    t.stage_rows = num("OVM_STAGE_ROWS", 0);
    t.acc_rows = num("OVM_ACC_ROWS", 0);
    t.stage_kib = (uint32_t)std::max(0, num("OVM_STAGE_KIB", 0));
    t.mle_points_in_lane = !off("OVM_MLE_POINTS");
    t.r0_batch = !off("OVM_R0_BATCH");
    t.mle_lds_slots = !off("OVM_MLE_LDS");
    t.mle_forward = !off("OVM_MLE_FWD");
    t.r0_groups_min = num("OVM_R0_GROUPS_MIN", -1);
    t.mle_batch_wgs = (uint32_t)std::max(0, num("OVM_MLE_BATCH", 256));
    t.mle_groups_max = (uint32_t)std::max(1, num("OVM_MLE_GROUPS_MAX", 128));
    t.gkr_three_points = getenv("OVM_GKR_THREE_POINTS") != nullptr;
    t.merkle_coop = !off("OVM_MERKLE_COOP");
    t.finish_inline_max = (uint32_t)std::max(1, num("OVM_FINISH_INLINE_MAX", 64));
    t.finish_atomic_max = (uint32_t)std::max(0, num("OVM_FINISH_ATOMIC_MAX", 0));
    t.download_kernel = !off("OVM_DOWNLOAD_KERNEL");
    t.early_lde = !off("OVM_EARLY_LDE");
    t.gkr_split_eq = !off("OVM_GKR_SPLIT_EQ");
    t.gkr_lookahead = num("OVM_GKR_LOOKAHEAD", 0) != 0;
    t.poseidon_lds_kib = (uint32_t)std::min(64, std::max(0, num("OVM_POSEIDON_LDS_KIB", 0)));
    t.gkr_la_max_log = (uint32_t)std::max(0, num("OVM_GKR_LA_MAX_LOG", 17));
    t.gkr_split_min_round = (uint32_t)std::max(3, num("OVM_GKR_SPLIT_MIN", 14));
    t.staged_forward = !off("OVM_STAGED_FWD");
    t.stacked_tail_device = std::max(0, std::min(2, num("OVM_STACKED_TAIL_DEVICE", 1)));
    t.mle_batch_lds_min_log = (uint32_t)std::max(0, num("OVM_MLE_BATCH_LDS_MIN_LOG", 9));
    t.lazy_w_accumulate = !off("OVM_LAZY_W");
    t.early_lde_lds_kib = (uint32_t)std::max(0, num("OVM_EARLY_LDE_LDS_KIB", 64));
    t.trace = getenv("OVM_TRACE") != nullptr;
    t.test_fail_rank = num("OVM_TEST_FAIL_RANK", -1);
    return t;
}

namespace {
// Validation at the C boundary: every violated precondition is an error code plus a message
// (ovm_hip_last_error), never a fault.
inline Fp checked_word(uint32_t w, const char *what) {
    if (w >= bb::P) throw OvmFail((int)hipErrorInvalidValue, std::string(what) + ": Montgomery word >= P");
    return Fp::raw(w);
}
// owner / rank (sharded proving): entries of traces held by another rank carry shape only
std::vector<TraceIn> traces_from_abi(const ProvingKey &pk, uint32_t n_traces, const ovm_trace_t *tr, bool need_data = true,
                                     const uint32_t *owner = nullptr, uint32_t rank = 0) {
    if (!tr) throw OvmFail((int)hipErrorInvalidValue, "traces");
    std::vector<TraceIn> in;
    std::vector<char> seen(pk.airs.size(), 0);
    for (uint32_t t = 0; t < n_traces; t++) {
        TraceIn ti;
        ti.air_idx = tr[t].air_idx;
        if (ti.air_idx >= pk.airs.size()) throw OvmFail((int)hipErrorInvalidValue, "air index out of range");
        if (seen[ti.air_idx]) throw OvmFail((int)hipErrorInvalidValue, "duplicate air index");
        seen[ti.air_idx] = 1;
        const AirKey &air = pk.airs[ti.air_idx];
        ti.height = tr[t].height;
        if (ti.height == 0 || (ti.height & (ti.height - 1))) throw OvmFail((int)hipErrorInvalidValue, "trace height");
        ti.owner = owner ? owner[t] : 0;
        if (ti.owner != rank) {
            in.push_back(std::move(ti));
            continue;
        }
        ti.d_trace = (const Fp *)tr[t].d_common_main;
        if (need_data && !ti.d_trace && air.common_main_width) throw OvmFail((int)hipErrorInvalidValue, "d_common_main is null");
        if (tr[t].num_public_values && !tr[t].public_values) throw OvmFail((int)hipErrorInvalidValue, "public_values is null");
        for (uint32_t i = 0; i < tr[t].num_public_values; i++) ti.public_values.push_back(checked_word(tr[t].public_values[i], "public value"));
        for (uint32_t i = 0; i < tr[t].num_cached; i++) {
            if (!tr[t].cached || !tr[t].cached[i]) throw OvmFail((int)hipErrorInvalidValue, "cached main handle");
            ti.cached.push_back(reinterpret_cast<const PcsData *>(tr[t].cached[i]));
        }
        in.push_back(std::move(ti));
    }
    return in;
}
void sponge_in(Prover &pr, const ovm_sponge_t *sponge) {
    if (!sponge) return;
    if (sponge->absorb_idx >= 8 || sponge->sample_idx > 8) throw OvmFail((int)hipErrorInvalidValue, "sponge indices");
    for (int i = 0; i < 16; i++) pr.ts.s.state[i] = checked_word(sponge->state[i], "sponge state");
    pr.ts.s.absorb_idx = sponge->absorb_idx;
    pr.ts.s.sample_idx = sponge->sample_idx;
}
void sponge_out(const Prover &pr, ovm_sponge_t *sponge) {
    if (!sponge) return;
    for (int i = 0; i < 16; i++) sponge->state[i] = pr.ts.s.state[i].v;
    sponge->absorb_idx = pr.ts.s.absorb_idx;
    sponge->sample_idx = pr.ts.s.sample_idx;
}
template <class F>
int guarded(F &&f, bool needs_device = true) {
    try {
        Tuning::refresh();  // (test processes only, OVM_TUNING_LIVE=1: the overrides are re-read here, once per call, and nowhere else)
        if (needs_device && !twiddles()) return (int)hipErrorInitializationError;
        const int rc = f();
        flush_uploads(nullptr);  // (nothing staged is left behind for a later call to trip over)
        return rc;
    } catch (const OvmFail &e) {
        pending_uploads().jobs.clear();
        g_last_error = e.what();
        return e.code ? e.code : (int)hipErrorUnknown;
    } catch (const std::exception &e) {
        pending_uploads().jobs.clear();
        g_last_error = e.what();
        return (int)hipErrorUnknown;
    }
}
}  // namespace


extern "C" {

int ovm_hip_pk_load(const uint32_t *blob, uint64_t words, ovm_pk_t **pk_out, ovm_stream_t stream) {
    if (!blob || !pk_out) return (int)hipErrorInvalidValue;
    return guarded([&]() -> int {
        auto dk = std::make_unique<DeviceKey>();
        dk->pk = parse_pk_blob(blob, words);
        dk->pre.resize(dk->pk.airs.size());
        // (a key without preprocessed traces is host data only: it loads without a device, which is what
        // ovm_hip_proof_size and the host-side self-tests need)
        bool any_pre = false;
        for (auto &a : dk->pk.airs) any_pre |= a.preprocessed_width != 0;
        if (any_pre && !twiddles()) return (int)hipErrorInitializationError;
        // keygen's part of the device work: upload and commit every preprocessed trace on its own
        // (keygen/mod.rs:283-300; DeviceDataTransporter::transport_pk_to_device, hal.rs:141-150)
        for (size_t a = 0; a < dk->pk.airs.size(); a++) {
            const AirKey &air = dk->pk.airs[a];
            if (!air.preprocessed_width) continue;
            Prover pr(*dk, (hipStream_t)stream);
            DevBuf<Fp> dev(air.preprocessed_vals.size());
            OVM_THROW_IF(hipMemcpyAsync(dev.p, air.preprocessed_vals.data(), air.preprocessed_vals.size() * sizeof(Fp),
                                        hipMemcpyHostToDevice, (hipStream_t)stream));
            auto pcs = std::make_unique<PcsData>(
                pr.commit_mats({{dev.p, air.preprocessed_width, air.preprocessed_height}}));
            pr.cx.sync();
            pcs->own_mat = std::move(dev);
            dk->pre[a] = std::move(pcs);
        }
        *pk_out = reinterpret_cast<ovm_pk_t *>(dk.release());
        return 0;
    }, false);
}

void ovm_hip_pk_free(ovm_pk_t *pk) { delete reinterpret_cast<DeviceKey *>(pk); }

int ovm_hip_pk_preprocessed_commitment(const ovm_pk_t *pk_, uint32_t air_idx, uint32_t commitment_out[8]) {
    if (!pk_ || !commitment_out) return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    if (air_idx >= dk.pre.size() || !dk.pre[air_idx]) return (int)hipErrorInvalidValue;
    for (int i = 0; i < 8; i++) commitment_out[i] = dk.pre[air_idx]->root.d[i].v;
    return 0;
}

// The key's own device copy of AIR air_idx's preprocessed trace and its PCS data, BORROWED (valid while the key lives;
// never freed by the caller): what `DeviceStarkProvingKey::preprocessed_data` hands out, without committing the trace
// a second time.
int ovm_hip_pk_preprocessed_data(const ovm_pk_t *pk_, uint32_t air_idx, const uint32_t **d_trace_out, uint64_t *width_out,
                                 uint64_t *height_out, uint32_t commitment_out[8], ovm_pcs_t **pcs_out) {
    if (!pk_ || !d_trace_out || !width_out || !height_out || !commitment_out || !pcs_out) return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    if (air_idx >= dk.pre.size() || !dk.pre[air_idx]) return (int)hipErrorInvalidValue;
    PcsData *pcs = dk.pre[air_idx].get();
    *d_trace_out = (const uint32_t *)pcs->own_mat.p;
    *width_out = dk.pk.airs[air_idx].preprocessed_width;
    *height_out = dk.pk.airs[air_idx].preprocessed_height;
    for (int i = 0; i < 8; i++) commitment_out[i] = pcs->root.d[i].v;
    *pcs_out = reinterpret_cast<ovm_pcs_t *>(pcs);
    return 0;
}

int ovm_hip_set_prover_config(const ovm_prover_config_t *cfg) {
    if (!cfg || cfg->round0_slab_log_rows > 30) return (int)hipErrorInvalidValue;
    ProverConfig &c = prover_config();
    c.cache_rs_code_matrix = cfg->cache_rs_code_matrix != 0;
    c.zerocheck_save_memory = cfg->zerocheck_save_memory != 0;
    if (cfg->round0_slab_log_rows) c.round0_slab_log_rows = cfg->round0_slab_log_rows;
    c.gkr_drop_leaves = cfg->gkr_drop_leaves != 0;
    block_cache().set_limit(cfg->cache_limit_bytes);
    return 0;
}

int ovm_hip_memory_stats(uint64_t *live_bytes, uint64_t *peak_live_bytes, uint64_t *cached_bytes, int reset_peak) {
    const uint64_t cached = block_cache().cached_bytes();
    PoolRegistry &reg = PoolRegistry::get();
    std::lock_guard<std::mutex> lk(reg.mu);
    if (live_bytes) *live_bytes = reg.live_bytes;
    if (peak_live_bytes) *peak_live_bytes = reg.peak_live_bytes;
    if (cached_bytes) *cached_bytes = cached;
    if (reset_peak) reg.peak_live_bytes = reg.live_bytes, reg.peak_segment_bytes = reg.segment_bytes;
    return 0;
}

int ovm_hip_pool_stats(uint64_t out4[4]) {
    if (!out4) return (int)hipErrorInvalidValue;
    PoolRegistry &reg = PoolRegistry::get();
    std::lock_guard<std::mutex> lk(reg.mu);
    out4[0] = reg.segment_bytes;
    out4[1] = reg.peak_segment_bytes;
    out4[2] = reg.driver_allocs;
    out4[3] = reg.owner.size();
    return 0;
}

uint64_t ovm_hip_constraint_eval_buffer_size(const ovm_pk_t *pk_, uint32_t air_idx) {
    if (!pk_) return 0;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    if (air_idx >= dk.pk.airs.size()) return 0;
    const AirKey &a = dk.pk.airs[air_idx];
    const uint64_t cdeg = a.max_constraint_degree;
    const uint64_t n_cosets = std::max<uint64_t>(cdeg ? cdeg - 1 : 0, a.interactions.empty() ? 0 : cdeg);
    return n_cosets * column_map_of(a).total * sizeof(uint32_t);
}

int ovm_hip_trim(void) {
    block_cache().trim();
    return 0;
}

// Host-only self-test of the staged program form (program.h: stage_program / staged_host_eval): for
// every AIR of the key, the zerocheck, LogUp and GKR-leaf programs are staged with the given layout
// and run on one random row (random public values and weights) next to the plain interpreter.
int ovm_hip_staged_selftest(const uint32_t *blob, uint64_t words, uint32_t stage_rows, uint32_t acc_rows, uint64_t seed,
                            uint64_t *blocks_out) {
    if (!blob || stage_rows < 2 || acc_rows < 1) return (int)hipErrorInvalidValue;
    return guarded([&]() -> int {
        const ProvingKey pk = parse_pk_blob(blob, words);
        uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1, n_blocks = 0;
        auto rnd = [&]() {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return Fp::from_canonical((uint32_t)((z ^ (z >> 31)) % bb::P));
        };
        auto rnd_ef = [&]() {
            EF e;
            for (int k = 0; k < 4; k++) e.c[k] = rnd();
            return e;
        };
        StagedLayout layout;
        layout.stage_rows = stage_rows;
        layout.acc_rows = acc_rows;
        for (size_t a = 0; a < pk.airs.size(); a++) {
            const AirKey &air = pk.airs[a];
            const ColumnMap cm = column_map_of(air);
            std::vector<Fp> row(cm.total), pv(air.num_public_values);
            for (auto &x : row) x = rnd();
            for (auto &x : pv) x = rnd();
            auto fetch = [&](uint32_t col) { return row[col]; };
            auto fail = [&](const char *what) {
                throw OvmFail(1, "staged self-test: AIR " + std::to_string(a) + " " + what + " program differs");
            };
            for (int kind = 0; kind < 2; kind++) {  // ACC programs: zerocheck, LogUp
                if (kind == 1 && air.interactions.empty()) continue;
                uint32_t nw = (uint32_t)air.constraint_idx.size();
                std::vector<Root> roots = kind == 0 ? roots_zerocheck(air) : roots_logup(air, &nw);
                std::vector<EF> w(std::max<uint32_t>(nw, 1));
                for (auto &x : w) x = rnd_ef();
                const Program p = compile_program(air, cm, pv, roots, nw);
                EF want[2] = {EF::zero(), EF::zero()}, got[2] = {EF::zero(), EF::zero()};
                std::vector<Fp> slots(p.num_slots);
                run_program_acc<Fp>(p.code.data(), (uint32_t)p.code.size(), w.data(), slots.data(), fetch, want);
                {   // independent of the compiler: every DAG node in index order, then sum_r W[r] * value(root r)
                    std::vector<Fp> val(air.nodes.size());
                    for (size_t i = 0; i < air.nodes.size(); i++) {
                        const Node &nd = air.nodes[i];
                        switch (nd.kind) {
                            case N_VAR:
                                val[i] = nd.entry == E_PUBLIC ? pv.at(nd.index)
                                         : nd.entry == E_MAIN ? row[cm.col(nd.part_index, nd.offset, nd.index)]
                                                              : row[cm.pre_col(nd.offset, nd.index)];
                                break;
                            case N_IS_FIRST: val[i] = row[0]; break;
                            case N_IS_TRANSITION: val[i] = row[1]; break;
                            case N_IS_LAST: val[i] = row[2]; break;
                            case N_CONST: val[i] = nd.constant; break;
                            case N_ADD: val[i] = val[nd.left] + val[nd.right]; break;
                            case N_SUB: val[i] = val[nd.left] - val[nd.right]; break;
                            case N_MUL: val[i] = val[nd.left] * val[nd.right]; break;
                            default: val[i] = -val[nd.left];
                        }
                    }
                    EF naive[2] = {EF::zero(), EF::zero()};
                    for (auto &r : roots) naive[r.acc] = naive[r.acc] + w[r.weight] * val[r.node];
                    if (!(naive[0] == want[0]) || !(naive[1] == want[1])) fail(kind == 0 ? "zerocheck (compiled vs DAG)" : "LogUp (compiled vs DAG)");
                }
                // the recomputation ladder and the grouped form give the same sums
                for (uint32_t limit : {1u, 3u, 1000u}) {
                    const Program q = compile_program_with(air, cm, pv, roots, nw, limit);
                    EF alt[2] = {EF::zero(), EF::zero()};
                    std::vector<Fp> sl(q.num_slots);
                    run_program_acc<Fp>(q.code.data(), (uint32_t)q.code.size(), w.data(), sl.data(), fetch, alt);
                    if (!(alt[0] == want[0]) || !(alt[1] == want[1])) fail("recomputing");
                }
                const StagedProgram sp = stage_program(p, w, layout);
                n_blocks += sp.blocks.size();
                staged_host_eval(sp, row, got, [&](uint32_t, Fp, const EF &) { fail("ACC (unexpected LEAF)"); });
                if (!(got[0] == want[0]) || !(got[1] == want[1])) fail(kind == 0 ? "zerocheck" : "LogUp");
            }
            if (!air.interactions.empty()) {  // DEN/LEAF program
                size_t max_len = 0;
                for (auto &it : air.interactions) max_len = std::max(max_len, it.message.size());
                std::vector<EF> w(max_len + 1), lc(air.interactions.size());
                for (auto &x : w) x = rnd_ef();
                for (auto &x : lc) x = rnd_ef();
                const Program p = compile_program(air, cm, pv, roots_leaves(air), (uint32_t)w.size());
                std::vector<std::pair<Fp, EF>> want(air.interactions.size()), got(air.interactions.size());
                std::vector<Fp> slots(p.num_slots);
                auto emit_want = [&](uint32_t k, Fp numer, const EF &den) { want.at(k) = {numer, den}; };
                run_program_leaves<Fp>(p.code.data(), (uint32_t)p.code.size(), w.data(), lc.data(), slots.data(), fetch, emit_want);
                const StagedProgram sp = stage_program(p, w, layout);
                n_blocks += sp.blocks.size();
                EF unused[2] = {EF::zero(), EF::zero()};
                staged_host_eval(sp, row, unused, [&](uint32_t k, Fp numer, const EF &den) { got.at(k) = {numer, den + lc.at(k)}; });
                for (size_t k = 0; k < want.size(); k++)
                    if (!(got[k].first.v == want[k].first.v) || !(got[k].second == want[k].second)) fail("GKR-leaf");
            }
        }
        if (blocks_out) *blocks_out = n_blocks;
        return 0;
    }, false);
}

// Size of the compiled register program of one AIR (kind 0: zerocheck, 1: LogUp round sums, 2: GKR leaves).
int ovm_hip_program_stats(const uint32_t *blob, uint64_t words, uint32_t air_idx, uint32_t kind, uint32_t *instrs_out,
                          uint32_t *slots_out, uint32_t *remat_limit_out) {
    if (!blob || kind > 2) return (int)hipErrorInvalidValue;
    return guarded([&]() -> int {
        const ProvingKey pk = parse_pk_blob(blob, words);
        if (air_idx >= pk.airs.size()) throw OvmFail(1, "program stats: no such AIR");
        const AirKey &air = pk.airs[air_idx];
        const std::vector<Fp> pv(air.num_public_values, Fp::zero());
        const std::vector<Root> roots = kind == 0 ? roots_zerocheck(air) : kind == 1 ? roots_logup(air) : roots_leaves(air);
        CompileStats st;
        const Program p = compile_program(air, column_map_of(air), pv, roots, 0, PROGRAM_SLOT_TARGET, &st);
        if (instrs_out) *instrs_out = (uint32_t)p.code.size();
        if (slots_out) *slots_out = p.num_slots;
        if (remat_limit_out) *remat_limit_out = st.remat_limit;
        return 0;
    }, false);
}

const char *ovm_hip_last_error(void) { return g_last_error.c_str(); }

int ovm_hip_prove(const ovm_pk_t *pk_, uint32_t n_traces, const ovm_trace_t *traces, ovm_sponge_t *sponge,
                  uint8_t *proof_out, uint64_t proof_cap, uint64_t *proof_len, uint64_t *section_offsets5,
                  ovm_stream_t stream) {
    if (!pk_ || !proof_len || n_traces == 0 || !traces) return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    return guarded([&]() -> int {
        // the proof length is a function of the shapes: a buffer that is too small is refused BEFORE any work
        {
            std::vector<ShapeIn> sh;
            for (uint32_t t = 0; t < n_traces; t++) sh.push_back({traces[t].air_idx, (size_t)traces[t].height});
            const uint64_t need = proof_section_ends(dk.pk, sh)[4];
            *proof_len = need;
            if (need > proof_cap || !proof_out)
                throw OvmFail((int)hipErrorInvalidValue, "proof buffer too small: " + std::to_string(need) + " bytes needed");
        }
        Prover pr(dk, (hipStream_t)stream);
        sponge_in(pr, sponge);
        pr.prove(traces_from_abi(dk.pk, n_traces, traces));
        if (pr.out.buf.size() != *proof_len) throw OvmFail((int)hipErrorUnknown, "internal: proof length differs from ovm_hip_proof_size");
        if (section_offsets5)
            for (size_t i = 0; i < 5 && i < pr.section_offsets.size(); i++) section_offsets5[i] = pr.section_offsets[i];
        sponge_out(pr, sponge);
        memcpy(proof_out, pr.out.buf.data(), pr.out.buf.size());
        return 0;
    });
}

int ovm_hip_prove_sharded(const ovm_pk_t *pk_, const ovm_shard_t *shard, uint32_t n_traces, const ovm_trace_t *traces,
                          const uint32_t *owner, ovm_sponge_t *sponge, uint8_t *proof_out, uint64_t proof_cap,
                          uint64_t *proof_len, uint64_t *section_offsets5, ovm_stream_t stream) {
    if (!pk_ || !proof_len || n_traces == 0 || !traces || !shard || !owner || shard->world == 0 || shard->rank >= shard->world)
        return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    bool working = false;  // past the unanimous validation: a failure from here on leaves the peers waiting
    const int rc = guarded([&]() -> int {
        const bool root = shard->rank == 0;
        Prover pr(dk, (hipStream_t)stream);
        pr.sh = ShardComm(shard);
        // Validation: each rank checks what only it can see (rank 0 its proof buffer, an owner its trace pointers, public
        // values and cached handles), then ONE exchange of the results: either every rank starts working or every rank
        // refuses with the first failing rank's code -- nobody is left waiting, the communicator stays in step.
        int32_t mine_rc = 0;
        std::string mine_msg;
        uint64_t need = 0;
        try {
            std::vector<ShapeIn> sh;
            for (uint32_t t = 0; t < n_traces; t++) sh.push_back({traces[t].air_idx, (size_t)traces[t].height});
            need = proof_section_ends(dk.pk, sh)[4];
            *proof_len = root ? need : 0;
            if (root && (need > proof_cap || !proof_out))
                throw OvmFail((int)hipErrorInvalidValue, "proof buffer too small: " + std::to_string(need) + " bytes needed");
            sponge_in(pr, sponge);
            pr.set_traces(traces_from_abi(dk.pk, n_traces, traces, true, owner, shard->rank));
        } catch (const OvmFail &e) {
            mine_rc = e.code ? e.code : (int)hipErrorUnknown;
            mine_msg = e.what();
        }
        if (!pr.sh.single()) {
            std::vector<int32_t> all(pr.sh.world);
            pr.sh.allgather(&mine_rc, all.data(), sizeof(int32_t));
            for (uint32_t r = 0; r < pr.sh.world; r++)
                if (all[r] != 0)
                    throw OvmFail(all[r], r == shard->rank ? mine_msg
                                                           : "rank " + std::to_string(r) + " rejected its inputs (code " + std::to_string(all[r]) +
                                                                 "); no rank started working");
        } else if (mine_rc) {
            throw OvmFail(mine_rc, mine_msg);
        }
        working = true;
        pr.prove_loaded();
        // the transcript is replicated: every rank hands its caller the same final sponge state
        sponge_out(pr, sponge);
        if (root) {
            if (pr.out.buf.size() != need) throw OvmFail((int)hipErrorUnknown, "internal: proof length differs from ovm_hip_proof_size");
            if (section_offsets5)
                for (size_t i = 0; i < 5 && i < pr.section_offsets.size(); i++) section_offsets5[i] = pr.section_offsets[i];
            memcpy(proof_out, pr.out.buf.data(), pr.out.buf.size());
        }
        return 0;
    });
    // This rank failed while the others work on: tell them (they return OVM_ESHARD_PEER_ABORTED instead of waiting for
    // the time-out) -- unless the failure IS a peer's abort or the exchange's own verdict, which every rank already has.
    // A TIME-OUT is announced whether or not work had started: it poisons only the rank that timed out, so without the
    // abort word the peers of a rank that gave up in the validation exchange would spin for their own full time-out.
    if (rc != 0 && (working || rc == OVM_ESHARD_TIMEOUT) && shard->world > 1 && shard->abort && rc != OVM_ESHARD_PEER_ABORTED &&
        rc != OVM_ESHARD_POISONED)
        (void)shard->abort(shard->ctx, rc);
    return rc;
}

int ovm_hip_commit(const ovm_pk_t *pk_, uint32_t n_traces, const ovm_trace_t *traces, uint32_t commitment_out[8],
                   ovm_pcs_t **pcs_out, ovm_stream_t stream) {
    if (!pk_ || !pcs_out || !commitment_out || n_traces == 0 || !traces) return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    return guarded([&]() -> int {
        Prover pr(dk, (hipStream_t)stream);
        pr.set_traces(traces_from_abi(dk.pk, n_traces, traces), false);
        auto *pcs = new PcsData(pr.commit());
        pr.cx.sync();
        for (int i = 0; i < 8; i++) commitment_out[i] = pcs->root.d[i].v;
        *pcs_out = reinterpret_cast<ovm_pcs_t *>(pcs);
        return 0;
    });
}

int ovm_hip_commit_matrix(const ovm_pk_t *pk_, const uint32_t *d_matrix, uint64_t width, uint64_t height,
                          uint32_t commitment_out[8], ovm_pcs_t **pcs_out, ovm_stream_t stream) {
    if (!pk_ || !pcs_out || !commitment_out || !d_matrix || width == 0 || height == 0 || (height & (height - 1)))
        return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    return guarded([&]() -> int {
        Prover pr(dk, (hipStream_t)stream);
        auto *pcs = new PcsData(pr.commit_mats({{(const Fp *)d_matrix, (size_t)width, (size_t)height}}));
        pr.cx.sync();
        for (int i = 0; i < 8; i++) commitment_out[i] = pcs->root.d[i].v;
        *pcs_out = reinterpret_cast<ovm_pcs_t *>(pcs);
        return 0;
    });
}

int ovm_hip_commit_matrices(const ovm_pcs_params_t *params, uint32_t n, const ovm_matrix_t *mats, uint32_t commitment_out[8],
                            ovm_pcs_t **pcs_out, ovm_stream_t stream) {
    if (!params || !mats || n == 0 || !commitment_out || !pcs_out) return (int)hipErrorInvalidValue;
    return guarded([&]() -> int {
        DeviceKey dk;  // only the PCS parameters of a key matter to a commitment
        dk.pk.params.l_skip = params->l_skip;
        dk.pk.params.n_stack = params->n_stack;
        dk.pk.params.log_blowup = params->log_blowup;
        dk.pk.params.k_whir = params->k_whir;
        check_pcs_params(dk.pk.params);
        std::vector<Prover::MatIn> in;
        for (uint32_t i = 0; i < n; i++) {
            const ovm_matrix_t &m = mats[i];
            if (m.height == 0 || (m.height & (m.height - 1))) throw OvmFail((int)hipErrorInvalidValue, "matrix height");
            if (m.width && !m.d_ptr) throw OvmFail((int)hipErrorInvalidValue, "matrix pointer is null");
            if (i && m.height > mats[i - 1].height)  // (stacked_pcs.rs:294: "must already be in height-sorted order")
                throw OvmFail((int)hipErrorInvalidValue, "matrices must be sorted by height, descending");
            in.push_back({(const Fp *)m.d_ptr, (size_t)m.width, (size_t)m.height});
        }
        Prover pr(dk, (hipStream_t)stream);
        auto *pcs = new PcsData(pr.commit_mats(in));
        pr.cx.sync();
        for (int i = 0; i < 8; i++) commitment_out[i] = pcs->root.d[i].v;
        *pcs_out = reinterpret_cast<ovm_pcs_t *>(pcs);
        return 0;
    });
}

int ovm_hip_proof_size(const ovm_pk_t *pk_, uint32_t n_traces, const ovm_trace_t *traces, uint64_t section_ends5[5]) {
    if (!pk_ || !traces || n_traces == 0 || !section_ends5) return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    return guarded([&]() -> int {
        std::vector<ShapeIn> sh;
        for (uint32_t t = 0; t < n_traces; t++) sh.push_back({traces[t].air_idx, (size_t)traces[t].height});
        const auto ends = proof_section_ends(dk.pk, sh);
        for (int i = 0; i < 5; i++) section_ends5[i] = ends[i];
        return 0;
    }, false);
}

void ovm_hip_pcs_free(ovm_pcs_t *pcs) { delete reinterpret_cast<PcsData *>(pcs); }

int ovm_hip_prove_rap_constraints(const ovm_pk_t *pk_, uint32_t n_traces, const ovm_trace_t *traces,
                                  ovm_sponge_t *sponge, uint8_t *out, uint64_t out_cap, uint64_t *gkr_len,
                                  uint64_t *total_len, uint32_t *r_out, uint32_t r_cap, uint32_t *r_len,
                                  ovm_stream_t stream) {
    if (!pk_ || !sponge || !total_len || !r_len || n_traces == 0 || !traces) return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    return guarded([&]() -> int {
        Prover pr(dk, (hipStream_t)stream);
        sponge_in(pr, sponge);
        pr.set_traces(traces_from_abi(dk.pk, n_traces, traces));
        std::vector<EF> r = pr.prove_rap_constraints();
        pr.cx.sync();
        if (gkr_len) *gkr_len = pr.section_offsets.empty() ? 0 : pr.section_offsets[0];
        *total_len = pr.out.buf.size();
        *r_len = (uint32_t)r.size();
        // too small: the sizes needed are in *total_len / *r_len and the caller's sponge is untouched, so the
        // call can be repeated (ovm_hip_proof_size gives the sizes beforehand)
        if (pr.out.buf.size() > out_cap || !out || r.size() > r_cap || !r_out)
            throw OvmFail((int)hipErrorInvalidValue, "output buffer too small");
        sponge_out(pr, sponge);
        memcpy(out, pr.out.buf.data(), pr.out.buf.size());
        for (size_t i = 0; i < r.size(); i++)
            for (int k = 0; k < 4; k++) r_out[4 * i + k] = r[i].c[k].v;
        return 0;
    });
}

int ovm_hip_prove_openings(const ovm_pk_t *pk_, uint32_t n_traces, const ovm_trace_t *traces, ovm_pcs_t *common_main,
                           const uint32_t *r, uint32_t r_len, ovm_sponge_t *sponge, uint8_t *out, uint64_t out_cap,
                           uint64_t *stacking_len, uint64_t *total_len, ovm_stream_t stream) {
    if (!pk_ || !sponge || !total_len || !common_main || !r || n_traces == 0 || !traces) return (int)hipErrorInvalidValue;
    const DeviceKey &dk = *reinterpret_cast<const DeviceKey *>(pk_);
    return guarded([&]() -> int {
        Prover pr(dk, (hipStream_t)stream);
        sponge_in(pr, sponge);
        pr.set_traces(traces_from_abi(dk.pk, n_traces, traces, false), false);
        std::vector<EF> rv(r_len);
        for (uint32_t i = 0; i < r_len; i++)
            for (int k = 0; k < 4; k++) rv[i].c[k] = checked_word(r[4 * i + k], "r");
        const PcsData *cm = reinterpret_cast<const PcsData *>(common_main);
        pr.prove_openings(cm, cm->root, rv);
        pr.cx.sync();
        if (stacking_len) *stacking_len = pr.section_offsets.empty() ? 0 : pr.section_offsets[0];
        *total_len = pr.out.buf.size();
        if (pr.out.buf.size() > out_cap || !out) throw OvmFail((int)hipErrorInvalidValue, "output buffer too small");
        sponge_out(pr, sponge);
        memcpy(out, pr.out.buf.data(), pr.out.buf.size());
        return 0;
    });
}

}  // extern "C"
