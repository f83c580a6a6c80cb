// Bulk device exchange of a sharded proof WITHOUT a collective library: every rank publishes an IPC
// handle of the allocation its send buffer lives in, the peers map it once (mappings are kept: the
// memory pool's segments persist across proofs) and PULL their part with plain device-to-device
// copies on their own stream -- over xGMI between the GPUs of a node, each pair on its own link, which is
// exactly the traffic pattern of the all-to-all + all-gather the prover asks for.
// The handles and the two barriers of an exchange travel through the shared-memory communicator
// (shard.cpp). Ranks that share one GPU (the tests) work the same way: IPC between processes on one
// device. The reference has no peer for this file; a deployment that prefers RCCL passes its own
// callbacks in ovm_shard_t instead (the Python host does, with this as its first fall-back).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "openvm_hip.h"

extern "C" int ovm_hip_shm_allgather(void *comm, const void *mine, void *all, uint64_t bytes);
extern "C" int ovm_hip_shm_comm_abort(void *comm, int code);

// ---- hipIpcOpenMemHandle that cannot hold a proof for ever --------------------------------------------------------
// Round 3 saw `hipIpcOpenMemHandle` of a peer's 3.2 GB segment not return when two 85 GB processes shared one GPU
// (DESIGN.md §9). Two defences, both on the MAPPING path only (a mapping is made once per peer segment and kept):
//  (1) opens are PHASED: rank 0 maps what it needs while every other rank waits in a barrier, then rank 1, ... -- no two
//      processes are ever inside the call at the same time. (With dmabuf IPC the importer asks the exporter PROCESS for
//      the buffer; two processes importing from each other at once is the one pattern the small-size runs that work and
//      the large-size run that hung differ in only by how long the call takes.)
//  (2) every open runs on a helper thread the caller waits for with a deadline (OVM_IPC_OPEN_TIMEOUT_S, default 20 s).
//      A call that misses it is abandoned (the thread is detached; if it ever returns, its mapping is closed), the
//      failure is agreed by all ranks, and the communicator DEGRADES for good: this and every later bulk exchange is
//      staged through pinned host memory and the shared-memory all-gather -- slow (a few GB/s), never stuck.
// Test hooks (the test twin of the library only, -DOVM_TEST_HOOKS=1): OVM_IPC_FORCE_OPEN_FAIL=1 (every open fails with
// hipErrorMapFailed; =2: with hipErrorInvalidValue, the code a stale handle or a non-hipMalloc buffer really produces),
// OVM_IPC_FORCE_OPEN_HANG=1 (the helper never returns), OVM_IPC_NO_BOUNCE=1 (export an allocation of an unsafe size as it is).
namespace {
struct OpenJob {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    hipError_t err = hipSuccess;
    void *p = nullptr;
};

hipError_t open_with_deadline(void **out, const hipIpcMemHandle_t &handle, double timeout_s,
                              int force /*0 none, 1 fail, 2 hang, 3 fail with hipErrorInvalidValue*/, bool &timed_out) {
    timed_out = false;
    if (force == 1) return hipErrorMapFailed;
    if (force == 3) return hipErrorInvalidValue;
    int device = 0;
    (void)hipGetDevice(&device);
    auto job = std::make_shared<OpenJob>();
    std::thread([job, handle, device, force]() {
        void *p = nullptr;
        hipError_t e = hipSetDevice(device);
        if (force == 2) std::this_thread::sleep_for(std::chrono::hours(24 * 365));
        if (e == hipSuccess) e = hipIpcOpenMemHandle(&p, handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) (void)hipGetLastError();
        std::unique_lock<std::mutex> lk(job->m);
        job->err = e;
        job->p = p;
        job->done = true;
        const bool abandoned = job->abandoned;
        lk.unlock();
        job->cv.notify_all();
        if (abandoned && e == hipSuccess && p) (void)hipIpcCloseMemHandle(p);  // nobody is waiting for it any more
    }).detach();
    std::unique_lock<std::mutex> lk(job->m);
    if (!job->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return job->done; })) {
        job->abandoned = true;
        timed_out = true;
        return hipErrorNotReady;
    }
    *out = job->p;
    return job->err;
}
}  // namespace

struct ovm_ipc_comm {
    ovm_shm_comm_t *shm = nullptr;
    uint32_t rank = 0, world = 1;
    // degraded: mapping a peer's buffer failed or timed out once (agreed by all ranks): bulk exchanges are staged
    // through host memory from then on. `why` says what happened (ovm_hip_ipc_comm_status).
    bool degraded = false;
    std::string why;
    double open_timeout_s = 20.0;
    int force_open = 0;
    uint64_t opens = 0, staged_exchanges = 0, bounced_exchanges = 0;
    // The runtime of this image (ROCm 7.2, dmabuf IPC) never returns from hipIpcOpenMemHandle when BIT 31 OF THE EXPORTED
    // ALLOCATION'S SIZE is set -- 2.0-3.99 GiB, 6.0-7.99 GiB, 11 GiB hang; 1.9, 4.0, 5.99, 12 GiB map in 2 ms, whatever else
    // the processes hold (tools/ipc_repro.py, profiles/r04/ipc_repro_sizes_r04.log). An owner whose send buffer lies in such
    // an allocation exports a COPY instead: a bounce buffer of its own with a safe size (one device-to-device copy per
    // exchange, ~0.1 ms per 256 MB; nothing to wait out, no host staging). The deadline stays for what this rule misses.
    uint8_t *d_bounce = nullptr;
    uint64_t d_bounce_cap = 0;
    bool no_bounce = false;  // (test hook: export the caller's allocation whatever its size -- the reproducer)
    // The size rule above was found on ROCm 7.2 and is a property of THAT runtime: re-validate it per runtime version
    // (tools/ipc_repro.py; the opt-in raw-hang test of tests/test_gpu_shard.py).
    static bool unsafe_size(uint64_t size) { return (size >> 31) & 1; }
    static uint64_t safe_size_for(uint64_t need) {
        uint64_t cap = (need + (2u << 20) - 1) & ~uint64_t((2u << 20) - 1);
        if (unsafe_size(cap)) cap = ((cap >> 32) + 1) << 32;
        return cap;
    }
    // a bounce buffer of at least `need` bytes whose allocation has a safe size; false: none to be had (export the original)
    bool ensure_bounce(uint64_t need) {
        if (d_bounce_cap >= need) return true;
        if (d_bounce) (void)hipFree(d_bounce);
        d_bounce = nullptr;
        d_bounce_cap = 0;
        const uint64_t cap = safe_size_for(need);
        void *p = nullptr;
        if (hipMalloc(&p, cap) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        void *base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p) != hipSuccess || base != p || unsafe_size(size)) {
            (void)hipGetLastError();
            (void)hipFree(p);
            return false;
        }
        d_bounce = (uint8_t *)p;
        d_bounce_cap = cap;
        return true;
    }
    uint8_t *h_send = nullptr, *h_recv = nullptr;  // pinned staging of the degraded mode
    uint64_t h_send_cap = 0, h_recv_cap = 0;
    // mappings of peers' allocations, by (owner rank, handle bytes), with the owner's BUFFER ID of the allocation and the
    // exchange that used them last. The buffer id is unique per allocation for the life of the owner's process: if the
    // owner returned a segment to the driver and a later allocation came back with the same handle bytes, the id
    // differs and the stale mapping is closed and replaced instead of being pulled through. A mapping nobody
    // has used for STALE_AFTER exchanges is closed as well (its owner may have freed the segment long ago).
    struct Mapping {
        void *p;
        uint64_t last_used;
        uint64_t buffer_id;
    };
    std::map<std::string, Mapping> opened;
    uint64_t exchanges = 0;
    static constexpr uint64_t STALE_AFTER = 64;

    struct Msg {
        hipIpcMemHandle_t handle;
        uint64_t offset;
        uint64_t bytes;  // of the sender's whole buffer (checked against what the receiver expects)
        uint64_t buffer_id;  // HIP_POINTER_ATTRIBUTE_BUFFER_ID of the allocation (0 = not available)
        int32_t status;
        int32_t pad;
    };

    int barrier() { return ovm_hip_shm_allgather(shm, nullptr, nullptr, 0); }
    // The ranks disagree on the size of the exchange: a caller bug, reported as such -- with a code of its own, because
    // hipErrorInvalidValue is also what hipIpcOpenMemHandle / hipIpcGetMemHandle / hipMemGetAddressRange return for a stale
    // handle or a buffer that is not a hipMalloc allocation, and THOSE must degrade to host staging.
    static constexpr int E_SIZE_MISMATCH = 0x4F564D53;  // (never a HIP error code)

    // every rank's send buffer, mapped into this process (own rank: the pointer itself)
    int publish_and_map(const void *d_send, uint64_t send_bytes, hipStream_t stream, std::vector<const uint8_t *> &src) {
        Msg m{};
        m.bytes = send_bytes;
        // what was enqueued before the exchange must have produced the data before a peer reads it
        hipError_t e = hipStreamSynchronize(stream);
        void *base = nullptr;
        size_t size = 0;
        if (e == hipSuccess) e = hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)d_send);
        const uint8_t *exported = (const uint8_t *)d_send;
        if (e == hipSuccess && world > 1 && unsafe_size(size) && !no_bounce && ensure_bounce(send_bytes)) {
            e = hipMemcpyAsync(d_bounce, d_send, send_bytes, hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            base = d_bounce;
            exported = d_bounce;
            bounced_exchanges++;
        }
        if (e == hipSuccess) {
            // (asked for every time: a segment that went back to the driver may have a successor at the same address)
            e = hipIpcGetMemHandle(&m.handle, base);
            m.offset = (uint64_t)(exported - (const uint8_t *)base);
            unsigned long long id = 0;
            if (hipPointerGetAttribute(&id, HIP_POINTER_ATTRIBUTE_BUFFER_ID, (hipDeviceptr_t)base) == hipSuccess) m.buffer_id = id;
            else (void)hipGetLastError();
        }
        if (e != hipSuccess) (void)hipGetLastError();
        m.status = (int32_t)e;
        std::vector<Msg> all(world);
        // (this all-gather is also the barrier "every rank's data is ready")
        const int rc = ovm_hip_shm_allgather(shm, &m, all.data(), sizeof(Msg));
        if (rc) return rc;
        for (uint32_t r = 0; r < world; r++)
            if (all[r].status != 0) {  // every rank sees the same failure
                if (why.empty()) why = "rank " + std::to_string(r) + " could not export its buffer (code " + std::to_string(all[r].status) + ")";
                return all[r].status;
            }
        src.assign(world, nullptr);
        int first_error = 0;
        exchanges++;
        std::vector<std::string> keys(world);
        std::vector<uint32_t> to_open;
        for (uint32_t r = 0; r < world; r++) {
            if (all[r].bytes != send_bytes) first_error = first_error ? first_error : E_SIZE_MISMATCH;
            if (r == rank) {
                src[r] = (const uint8_t *)d_send;
                continue;
            }
            keys[r].assign((const char *)&all[r].handle, sizeof(hipIpcMemHandle_t));
            keys[r].append((const char *)&r, sizeof r);
            auto it = opened.find(keys[r]);
            if (it != opened.end() && it->second.buffer_id != all[r].buffer_id) {  // same handle bytes, another allocation
                (void)hipIpcCloseMemHandle(it->second.p);
                opened.erase(it);
                it = opened.end();
            }
            if (it == opened.end()) to_open.push_back(r);
        }
        // does ANY rank have something to map? (all ranks take the phased path together, or none does)
        std::vector<uint32_t> needs(world);
        const uint32_t mine_needs = (uint32_t)to_open.size();
        if (int rc2 = ovm_hip_shm_allgather(shm, &mine_needs, needs.data(), sizeof(uint32_t))) return rc2;
        bool any = false;
        for (uint32_t n : needs) any = any || n != 0;
        if (any) {
            for (uint32_t turn = 0; turn < world; turn++) {
                if (turn == rank) {
                    for (uint32_t r : to_open) {
                        void *p = nullptr;
                        bool timed_out = false;
                        const hipError_t oe = open_with_deadline(&p, all[r].handle, open_timeout_s, force_open, timed_out);
                        opens++;
                        if (oe != hipSuccess) {
                            (void)hipGetLastError();
                            first_error = first_error ? first_error : (int)oe;
                            if (why.empty())
                                why = timed_out ? "hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s buffer did not return within " +
                                                      std::to_string((int)open_timeout_s) + " s (OVM_IPC_OPEN_TIMEOUT_S)"
                                                : "hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s buffer failed with code " + std::to_string((int)oe);
                            break;  // (one failure degrades the communicator: no point in trying the other peers)
                        }
                        opened.emplace(keys[r], Mapping{p, exchanges, all[r].buffer_id});
                    }
                }
                // (the barrier of the turn carries "somebody's open failed": the communicator degrades anyway, and every
                // further open of a buffer that does not map could cost its rank the whole deadline again)
                const uint32_t failed = first_error != 0 && first_error != E_SIZE_MISMATCH;
                std::vector<uint32_t> fs(world);
                if (int rc2 = ovm_hip_shm_allgather(shm, &failed, fs.data(), sizeof(uint32_t))) return rc2;
                bool stop = false;
                for (uint32_t f : fs) stop = stop || f != 0;
                if (stop) {
                    if (!first_error) first_error = (int)hipErrorMapFailed;
                    break;
                }
            }
        }
        for (uint32_t r = 0; r < world; r++) {
            if (r == rank) continue;
            auto it = opened.find(keys[r]);
            if (it == opened.end()) {
                first_error = first_error ? first_error : (int)hipErrorMapFailed;
                continue;
            }
            it->second.last_used = exchanges;
            src[r] = (const uint8_t *)it->second.p + all[r].offset;
        }
        for (auto it = opened.begin(); it != opened.end();)
            if (exchanges - it->second.last_used > STALE_AFTER) {
                (void)hipIpcCloseMemHandle(it->second.p);
                it = opened.erase(it);
            } else {
                ++it;
            }
        return first_error;
    }

    // outcome agreed by all ranks (a rank that failed to map must not leave the others waiting in the pulls' barrier)
    int agree(int mine_rc) {
        std::vector<int32_t> all(world);
        const int32_t v = mine_rc;
        const int rc = ovm_hip_shm_allgather(shm, &v, all.data(), sizeof(int32_t));
        if (rc) return rc;
        for (int32_t x : all)
            if (x) return x;
        return 0;
    }

    // The shared-memory exchange's own verdicts (a peer died, aborted, the communicator is poisoned) are final; anything
    // else that went wrong while exporting or mapping degrades the communicator on every rank.
    static bool is_exchange_error(int rc) { return rc == OVM_ESHARD_TIMEOUT || rc == OVM_ESHARD_PEER_ABORTED || rc == OVM_ESHARD_POISONED; }
    // 0: mapped (src valid); 1: degraded (agreed by all ranks), stage this exchange; else: error
    int map_or_degrade(const void *d_send, uint64_t send_bytes, hipStream_t stream, std::vector<const uint8_t *> &src, int &err) {
        err = 0;
        if (degraded) return 1;
        const int mine = publish_and_map(d_send, send_bytes, stream, src);
        if (is_exchange_error(mine)) {
            err = mine;
            return 2;
        }
        const int rc = agree(mine);
        if (rc == 0) return 0;
        if (is_exchange_error(rc) || rc == E_SIZE_MISMATCH) {
            err = rc == E_SIZE_MISMATCH ? (int)hipErrorInvalidValue : rc;
            return 2;
        }
        degraded = true;
        if (why.empty()) why = "a peer could not map this rank's buffer (code " + std::to_string(rc) + ")";
        fprintf(stderr, "[ovm ipc] rank %u: %s -- bulk exchanges are STAGED THROUGH HOST MEMORY from now on\n", rank, why.c_str());
        return 1;
    }
    int ensure_host(uint8_t *&buf, uint64_t &cap, uint64_t need) {
        if (cap >= need) return 0;
        if (buf) (void)hipHostFree(buf);
        buf = nullptr;
        cap = 0;
        const hipError_t e = hipHostMalloc((void **)&buf, need, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return (int)e;
        }
        cap = need;
        return 0;
    }
    // degraded mode: device -> pinned host -> shared-memory all-gather -> device
    int staged_allgather(const void *d_mine, void *d_all, uint64_t bytes, hipStream_t stream) {
        staged_exchanges++;
        int rc = ensure_host(h_send, h_send_cap, bytes);
        if (!rc) rc = ensure_host(h_recv, h_recv_cap, bytes * world);
        hipError_t e = hipSuccess;
        if (!rc) {
            e = hipMemcpyAsync(h_send, d_mine, bytes, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) rc = (int)e;
        }
        rc = agree(rc);
        if (rc) return rc;
        if ((rc = ovm_hip_shm_allgather(shm, h_send, h_recv, bytes))) return rc;
        e = hipMemcpyAsync(d_all, h_recv, bytes * world, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) (void)hipGetLastError();
        return agree((int)e);
    }
    int staged_alltoall(const void *d_send, void *d_recv, uint64_t bytes, hipStream_t stream) {
        staged_exchanges++;
        int rc = ensure_host(h_send, h_send_cap, bytes * world);
        if (!rc) rc = ensure_host(h_recv, h_recv_cap, bytes * world);
        hipError_t e = hipSuccess;
        if (!rc) {
            e = hipMemcpyAsync(h_send, d_send, bytes * world, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) rc = (int)e;
        }
        rc = agree(rc);
        if (rc) return rc;
        // block j of every rank is gathered in turn; rank j keeps it (world x the bytes a pairwise exchange would move:
        // this is the fall-back that needs nothing, not the fast path)
        for (uint32_t j = 0; j < world; j++) {
            if ((rc = ovm_hip_shm_allgather(shm, h_send + (uint64_t)j * bytes, h_recv, bytes))) return rc;
            if (j == rank) {
                e = hipMemcpyAsync(d_recv, h_recv, bytes * world, hipMemcpyHostToDevice, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (e != hipSuccess) (void)hipGetLastError();
            }
        }
        return agree((int)e);
    }

    int allgather(const void *d_mine, void *d_all, uint64_t bytes, hipStream_t stream) {
        std::vector<const uint8_t *> src;
        int rc = 0;
        const int how = map_or_degrade(d_mine, bytes, stream, src, rc);
        if (how == 2) return rc;
        if (how == 1) return staged_allgather(d_mine, d_all, bytes, stream);
        hipError_t e = hipSuccess;
        for (uint32_t k = 0; k < world && e == hipSuccess; k++) {
            const uint32_t r = (rank + k) % world;  // start with the own block, then spread the peers
            e = hipMemcpyAsync((uint8_t *)d_all + (uint64_t)r * bytes, src[r], bytes, hipMemcpyDeviceToDevice, stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) (void)hipGetLastError();
        // nobody's send buffer may change before every rank has pulled from it
        rc = agree((int)e);
        return rc;
    }

    int alltoall(const void *d_send, void *d_recv, uint64_t bytes, hipStream_t stream) {
        std::vector<const uint8_t *> src;
        int rc = 0;
        const int how = map_or_degrade(d_send, bytes * world, stream, src, rc);
        if (how == 2) return rc;
        if (how == 1) return staged_alltoall(d_send, d_recv, bytes, stream);
        hipError_t e = hipSuccess;
        for (uint32_t k = 0; k < world && e == hipSuccess; k++) {
            const uint32_t s = (rank + k) % world;
            e = hipMemcpyAsync((uint8_t *)d_recv + (uint64_t)s * bytes, src[s] + (uint64_t)rank * bytes, bytes, hipMemcpyDeviceToDevice, stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) (void)hipGetLastError();
        return agree((int)e);
    }
};

extern "C" {

int ovm_hip_ipc_comm_create(ovm_shm_comm_t *shm, uint32_t rank, uint32_t world, ovm_ipc_comm_t **out) {
    if (!shm || !out || world == 0 || rank >= world) return (int)hipErrorInvalidValue;
    auto *c = new ovm_ipc_comm();
    c->shm = shm;
    c->rank = rank;
    c->world = world;
    if (const char *e = getenv("OVM_IPC_OPEN_TIMEOUT_S")) c->open_timeout_s = std::max(0.05, atof(e));
#if defined(OVM_TEST_HOOKS)
    if (const char *e = getenv("OVM_IPC_NO_BOUNCE")) c->no_bounce = atoi(e) != 0;
    if (const char *e = getenv("OVM_IPC_FORCE_OPEN_FAIL")) c->force_open = atoi(e) == 2 ? 3 : atoi(e) ? 1 : 0;
    if (const char *e = getenv("OVM_IPC_FORCE_OPEN_HANG")) c->force_open = atoi(e) ? 2 : c->force_open;
#endif
    *out = c;
    return 0;
}

void ovm_hip_ipc_comm_destroy(ovm_ipc_comm_t *c) {
    if (!c) return;
    for (auto &kv : c->opened) (void)hipIpcCloseMemHandle(kv.second.p);
    if (c->d_bounce) (void)hipFree(c->d_bounce);
    if (c->h_send) (void)hipHostFree(c->h_send);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    delete c;
}

int ovm_hip_ipc_comm_status(const ovm_ipc_comm_t *c, ovm_ipc_status_t *out) {
    if (!c || !out) return (int)hipErrorInvalidValue;
    memset(out, 0, sizeof *out);
    out->degraded = c->degraded ? 1 : 0;
    out->mappings = (uint32_t)c->opened.size();
    out->opens = c->opens;
    out->exchanges = c->exchanges;
    out->staged_exchanges = c->staged_exchanges;
    out->bounced_exchanges = c->bounced_exchanges;
    snprintf(out->reason, sizeof out->reason, "%s", c->why.c_str());
    return 0;
}

int ovm_hip_ipc_allgather_host(void *ctx, const void *mine, void *all, uint64_t bytes) {
    if (!ctx) return (int)hipErrorInvalidValue;
    return ovm_hip_shm_allgather(static_cast<ovm_ipc_comm *>(ctx)->shm, mine, all, bytes);
}

int ovm_hip_ipc_abort(void *ctx, int code) {
    if (!ctx) return (int)hipErrorInvalidValue;
    return ovm_hip_shm_comm_abort(static_cast<ovm_ipc_comm *>(ctx)->shm, code);
}

int ovm_hip_ipc_allgather_device(void *ctx, const void *d_mine, void *d_all, uint64_t bytes, ovm_stream_t stream) {
    if (!ctx || !d_mine || !d_all || bytes == 0) return (int)hipErrorInvalidValue;
    return static_cast<ovm_ipc_comm *>(ctx)->allgather(d_mine, d_all, bytes, (hipStream_t)stream);
}

int ovm_hip_ipc_alltoall_device(void *ctx, const void *d_send, void *d_recv, uint64_t bytes, ovm_stream_t stream) {
    if (!ctx || !d_send || !d_recv || bytes == 0) return (int)hipErrorInvalidValue;
    return static_cast<ovm_ipc_comm *>(ctx)->alltoall(d_send, d_recv, bytes, (hipStream_t)stream);
}

}  // extern "C"
