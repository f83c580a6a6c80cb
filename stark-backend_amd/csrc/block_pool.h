// Device memory pool of the prover (host code; HIP runtime calls only).
//
// What it is for: a proof allocates a few hundred buffers from kilobytes to tens of gigabytes, and
// hipMalloc / hipFree of multi-GB buffers cost hundreds of milliseconds and synchronise the device
// (config D: 990 ms per proof with the driver's allocator, 156 ms with round 1's first cache). So memory
// is taken from the driver in SEGMENTS that are kept, and buffers are BLOCKS of segments:
//   * a request takes the smallest free block that fits (best fit); a block noticeably larger than the
//     request is split and the remainder stays free; a freed block merges with free neighbours. Back-to-back
//     proofs of one shape replay the same split sequence over the same segments: no driver call after the
//     first proof. Proofs of DIFFERENT shapes (the 209-segment replay) reuse each other's
//     segments: round 2's first cache kept whole blocks only and ended that sweep holding 165 GiB for a
//     30 GiB peak.
//   * two size classes, as every caching allocator ends up with: requests below 1 MiB live in 8 MiB segments
//     of their own, so a long-lived 4 KB buffer never pins the head of a 3 GB block; a large request never
//     splits a block more than 8 times its size (a 16 MB request does not nibble a 27 GB codeword block).
//   * reuse is in stream order by construction: a pool belongs to one (thread, device) and all work of a
//     prover thread runs on one stream. A block freed by ANOTHER thread (a key or PCS handle dropped
//     elsewhere) goes back to its owner's pool after a device synchronisation.
//   * `limit` caps the free bytes kept: beyond it, fully free segments go back to the driver. A failed
//     hipMalloc releases every fully free segment and retries.
// This is the role the reference gives its VPMM pool (crates/cuda-common/src/memory_manager/,
// docs/vpmm_spec.md), which defragments by remapping physical pages under a reserved virtual range;
// merging neighbours inside kept segments gets most of that without the driver's VMM API, and 288 GB
// of HBM leaves room for the rest. `ovm_hip_trim` / `ovm_hip_memory_stats` are the controls.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <stdexcept>
#include <unordered_map>
#include <vector>

namespace ovm {

class BlockPool;

// Process-wide: which pool a live block belongs to (a block may be freed on any thread) and the totals
// ovm_hip_memory_stats reports.
struct PoolRegistry {
    std::mutex mu;
    std::unordered_map<void *, BlockPool *> owner;
    size_t live_bytes = 0, peak_live_bytes = 0;  // blocks handed out
    size_t segment_bytes = 0, peak_segment_bytes = 0;  // memory held from the driver
    uint64_t driver_allocs = 0;                   // hipMalloc calls ever made
    static PoolRegistry &get() {
        static PoolRegistry *r = new PoolRegistry();  // (leaked on purpose: thread-exit destructors may run late)
        return *r;
    }
};

// Where segments come from: the HIP runtime, or plain host memory in the device-free self-test
// (ovm_hip_pool_selftest).
struct PoolDriver {
    int (*alloc)(void **p, size_t bytes);
    void (*release)(int device, void *p);
    void (*sync)(int device);
    int device = 0;  // the device the pool's memory lives on (a block may come back on a thread whose current device is another)
    // the calling thread's current device for the duration of a call on `device`
    struct OnDevice {
        int saved = -1;
        explicit OnDevice(int device) {
            int cur = 0;
            if (hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess) saved = cur;
        }
        ~OnDevice() {
            if (saved >= 0) (void)hipSetDevice(saved);
        }
    };
    static PoolDriver hip() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return PoolDriver{[](void **p, size_t bytes) {  // (new segments are only ever taken by the owning thread, on its device)
                              const hipError_t e = hipMalloc(p, bytes);
                              if (e != hipSuccess) (void)hipGetLastError();
                              return (int)e;
                          },
                          [](int device, void *p) {
                              OnDevice g(device);
                              (void)hipFree(p);
                          },
                          [](int device) {
                              OnDevice g(device);
                              (void)hipDeviceSynchronize();
                          },
                          dev};
    }
};

class BlockPool {
  public:
    explicit BlockPool(PoolDriver drv = PoolDriver::hip()) : drv_(drv) {}
    static constexpr size_t SMALL_MAX = size_t(1) << 20;      // requests below this: the small class
    static constexpr size_t SMALL_SEGMENT = size_t(8) << 20;  // segment size of the small class
    static constexpr size_t LARGE_ROUND = size_t(2) << 20;    // large requests are multiples of this
    static constexpr size_t MAX_SPLIT_RATIO = 8;              // a large request splits blocks up to this many times its size

    struct Error : std::runtime_error {
        int code;
        Error(int c, const char *what) : std::runtime_error(what), code(c) {}
    };

    void *get(size_t bytes) {
        const bool small = bytes < SMALL_MAX;
        bytes = small ? round_up(std::max<size_t>(bytes, 1), 512) : round_up(bytes, LARGE_ROUND);
        Block *b = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu_);
            FreeMap &fm = small ? free_small_ : free_large_;
            auto it = fm.lower_bound(bytes);
            if (it != fm.end() && (small || it->first <= bytes * MAX_SPLIT_RATIO)) {
                b = it->second;
                fm.erase(it);
                cached_ -= b->size;
                b->free = false;
                // keep the tail free when it is worth a block of its own
                const size_t min_tail = small ? 512 : std::max(LARGE_ROUND, bytes / 8);
                if (b->size >= bytes + min_tail) {
                    Block *t = new Block{b->p + bytes, b->size - bytes, true, small, b, b->next, b->seg};
                    if (b->next) b->next->prev = t;
                    b->next = t;
                    b->size = bytes;
                    insert_free(t);
                }
            }
        }
        if (!b) b = new_segment(small ? std::max(bytes, SMALL_SEGMENT) : bytes, bytes, small);
        {
            std::lock_guard<std::mutex> lk(mu_);
            live_[b->p] = b;
            b->seg->live_blocks++;
        }
        PoolRegistry &reg = PoolRegistry::get();
        std::lock_guard<std::mutex> lk(reg.mu);
        reg.owner[b->p] = this;
        reg.live_bytes += b->size;
        reg.peak_live_bytes = std::max(reg.peak_live_bytes, reg.live_bytes);
        return b->p;
    }

    // Return a block from any thread. `caller` = the calling thread's pool for the block's device (may be null).
    static void put(void *p, BlockPool *caller) {
        BlockPool *pool = nullptr;
        {
            PoolRegistry &reg = PoolRegistry::get();
            std::lock_guard<std::mutex> lk(reg.mu);
            auto it = reg.owner.find(p);
            if (it == reg.owner.end()) return;  // not ours (or freed twice): nothing sane to do
            pool = it->second;
            reg.owner.erase(it);
        }
        // another thread's block: its stream knows nothing of what this thread's work did with the memory
        if (pool != caller) pool->drv_.sync(pool->drv_.device);
        if (pool->give_back(p)) delete pool;
    }

    // release every fully free segment to the driver
    void trim() {
        std::vector<void *> victims;
        {
            std::lock_guard<std::mutex> lk(mu_);
            collect_free_segments(victims, 0);
        }
        release(victims);
    }
    void set_limit(size_t bytes) {
        std::vector<void *> victims;
        {
            std::lock_guard<std::mutex> lk(mu_);
            limit_ = bytes;
            if (limit_ && cached_ > limit_) collect_free_segments(victims, limit_);
        }
        release(victims);
    }
    size_t cached_bytes() {
        std::lock_guard<std::mutex> lk(mu_);
        return cached_;
    }
    // the owning thread is exiting: free segments go now, the rest when their last block comes back
    // (returns true when the caller should delete the pool)
    bool orphan() {
        std::vector<void *> victims;
        {
            std::lock_guard<std::mutex> lk(mu_);
            orphaned_ = true;
            busy_++;
            collect_free_segments(victims, 0);
        }
        release(victims);
        return leave();
    }

  private:
    struct Segment {
        char *base;
        size_t size;
        size_t live_blocks;
    };
    struct Block {
        char *p;
        size_t size;
        bool free, small;
        Block *prev, *next;  // neighbours inside the segment, in address order
        Segment *seg;
        std::multimap<size_t, Block *>::iterator where;  // in the free map while free
    };
    using FreeMap = std::multimap<size_t, Block *>;

    PoolDriver drv_;
    std::mutex mu_;
    FreeMap free_small_, free_large_;
    std::unordered_map<void *, Block *> live_;
    size_t cached_ = 0, limit_ = 0;
    bool orphaned_ = false;
    // Calls that may still touch the pool after dropping mu_ (they release victims to the driver). Whether an orphaned,
    // empty pool may be deleted is decided by the LAST of them on its way out -- a give_back that saw "empty" while
    // another thread was still inside release() used to delete the pool under it.
    unsigned busy_ = 0;
    bool leave() {
        std::lock_guard<std::mutex> lk(mu_);
        return --busy_ == 0 && orphaned_ && live_.empty();
    }

    static size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
    void insert_free(Block *b) {  // (mu_ held)
        b->free = true;
        b->where = (b->small ? free_small_ : free_large_).emplace(b->size, b);
        cached_ += b->size;
    }
    void remove_free(Block *b) {  // (mu_ held)
        (b->small ? free_small_ : free_large_).erase(b->where);
        cached_ -= b->size;
    }

    // a new segment from the driver; returns its first block of `bytes` bytes (not free), the tail (if any) free
    Block *new_segment(size_t seg_bytes, size_t bytes, bool small) {
        void *base = nullptr;
        int e = drv_.alloc(&base, seg_bytes);
        if (e != 0) {
            trim();
            e = drv_.alloc(&base, seg_bytes);
        }
        if (e != 0) throw Error(e, "hipMalloc");
        {
            PoolRegistry &reg = PoolRegistry::get();
            std::lock_guard<std::mutex> lk(reg.mu);
            reg.driver_allocs++;
            reg.segment_bytes += seg_bytes;
            reg.peak_segment_bytes = std::max(reg.peak_segment_bytes, reg.segment_bytes);
        }
        Segment *seg = new Segment{(char *)base, seg_bytes, 0};
        Block *b = new Block{(char *)base, bytes, false, small, nullptr, nullptr, seg};
        if (seg_bytes > bytes) {
            std::lock_guard<std::mutex> lk(mu_);
            Block *t = new Block{(char *)base + bytes, seg_bytes - bytes, true, small, b, nullptr, seg};
            b->next = t;
            insert_free(t);
        }
        return b;
    }

    // the block comes back: merge with free neighbours; true = the pool is orphaned and now empty
    bool give_back(void *p) {
        std::vector<void *> victims;
        size_t size = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = live_.find(p);
            if (it == live_.end()) return false;
            busy_++;
            Block *b = it->second;
            live_.erase(it);
            size = b->size;
            b->seg->live_blocks--;
            if (b->next && b->next->free) {
                Block *n = b->next;
                remove_free(n);
                b->size += n->size;
                b->next = n->next;
                if (n->next) n->next->prev = b;
                delete n;
            }
            if (b->prev && b->prev->free) {
                Block *q = b->prev;
                remove_free(q);
                q->size += b->size;
                q->next = b->next;
                if (b->next) b->next->prev = q;
                delete b;
                b = q;
            }
            insert_free(b);
            if (orphaned_)
                collect_free_segments(victims, 0);
            else if (limit_ && cached_ > limit_)
                collect_free_segments(victims, limit_);
        }
        {
            PoolRegistry &reg = PoolRegistry::get();
            std::lock_guard<std::mutex> lk(reg.mu);
            reg.live_bytes -= size;
        }
        release(victims);
        return leave();
    }

    // (mu_ held) take fully free segments out of the pool, largest first, until at most `keep` free bytes remain
    void collect_free_segments(std::vector<void *> &victims, size_t keep) {
        for (FreeMap *fm : {&free_large_, &free_small_})
            for (auto it = fm->end(); it != fm->begin() && cached_ > keep;) {
                --it;
                Block *b = it->second;
                if (b->prev || b->next || b->size != b->seg->size) continue;  // part of a segment that is still in use
                victims.push_back(b->seg->base);
                seg_sizes_.push_back(b->seg->size);
                cached_ -= b->size;
                delete b->seg;
                delete b;
                it = fm->erase(it);
            }
    }
    std::vector<size_t> seg_sizes_;  // sizes of the victims collected (consumed by release)
    void release(std::vector<void *> &victims) {
        if (victims.empty()) return;
        size_t total = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t s : seg_sizes_) total += s;
            seg_sizes_.clear();
        }
        drv_.sync(drv_.device);
        for (void *v : victims) drv_.release(drv_.device, v);
        victims.clear();
        PoolRegistry &reg = PoolRegistry::get();
        std::lock_guard<std::mutex> lk(reg.mu);
        reg.segment_bytes -= total;
    }
};

}  // namespace ovm
