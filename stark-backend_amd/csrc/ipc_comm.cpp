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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "openvm_hip.h"

extern "C" int ovm_hip_shm_allgather(void *comm, const void *mine, void *all, uint64_t bytes);
extern "C" int ovm_hip_shm_comm_abort(void *comm, int code);

struct ovm_ipc_comm {
    ovm_shm_comm_t *shm = nullptr;
    uint32_t rank = 0, world = 1;
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

    // every rank's send buffer, mapped into this process (own rank: the pointer itself)
    int publish_and_map(const void *d_send, uint64_t send_bytes, hipStream_t stream, std::vector<const uint8_t *> &src) {
        Msg m{};
        m.bytes = send_bytes;
        // what was enqueued before the exchange must have produced the data before a peer reads it
        hipError_t e = hipStreamSynchronize(stream);
        void *base = nullptr;
        size_t size = 0;
        if (e == hipSuccess) e = hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)d_send);
        if (e == hipSuccess) {
            // (asked for every time: a segment that went back to the driver may have a successor at the same address)
            e = hipIpcGetMemHandle(&m.handle, base);
            m.offset = (uint64_t)((const uint8_t *)d_send - (const uint8_t *)base);
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
            if (all[r].status != 0) return all[r].status;  // every rank sees the same failure
        src.assign(world, nullptr);
        int first_error = 0;
        exchanges++;
        for (uint32_t r = 0; r < world; r++) {
            if (all[r].bytes != send_bytes) first_error = first_error ? first_error : (int)hipErrorInvalidValue;
            if (r == rank) {
                src[r] = (const uint8_t *)d_send;
                continue;
            }
            std::string key((const char *)&all[r].handle, sizeof(hipIpcMemHandle_t));
            key.append((const char *)&r, sizeof r);
            auto it = opened.find(key);
            if (it != opened.end() && it->second.buffer_id != all[r].buffer_id) {  // same handle bytes, another allocation
                (void)hipIpcCloseMemHandle(it->second.p);
                opened.erase(it);
                it = opened.end();
            }
            if (it == opened.end()) {
                void *p = nullptr;
                const hipError_t oe = hipIpcOpenMemHandle(&p, all[r].handle, hipIpcMemLazyEnablePeerAccess);
                if (oe != hipSuccess) {
                    (void)hipGetLastError();
                    first_error = first_error ? first_error : (int)oe;
                    continue;
                }
                it = opened.emplace(key, Mapping{p, exchanges, all[r].buffer_id}).first;
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

    int allgather(const void *d_mine, void *d_all, uint64_t bytes, hipStream_t stream) {
        std::vector<const uint8_t *> src;
        int rc = agree(publish_and_map(d_mine, bytes, stream, src));
        if (rc) return rc;
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
        int rc = agree(publish_and_map(d_send, bytes * world, stream, src));
        if (rc) return rc;
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
    *out = c;
    return 0;
}

void ovm_hip_ipc_comm_destroy(ovm_ipc_comm_t *c) {
    if (!c) return;
    for (auto &kv : c->opened) (void)hipIpcCloseMemHandle(kv.second.p);
    delete c;
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
