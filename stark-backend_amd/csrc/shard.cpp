// Exchange layer of the sharded prover (see shard.h): ShardComm wrappers over the ovm_shard_t callbacks
// and the built-in shared-memory all-gather for the ranks of one node. Host only.
#include "shard.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "host.h"

namespace ovm {

void Unpacker::need(size_t k) const {
    if (at + k > n) throw OvmFail(1, "shard exchange: truncated message");
}

void ShardComm::allgather(const void *mine, void *all, uint64_t bytes) const {
    if (single()) {
        if (bytes && all != mine) memcpy(all, mine, bytes);
        return;
    }
    if (!abi->allgather_host) throw OvmFail(1, "ovm_shard_t: allgather_host is null");
    const int rc = abi->allgather_host(abi->ctx, mine, all, bytes);
    if (rc != 0) {
        const char *why = ovm_hip_shard_last_error();
        throw OvmFail(rc, std::string("shard exchange: allgather_host failed") + (why && *why ? std::string(": ") + why : std::string()));
    }
}

std::vector<std::vector<uint8_t>> ShardComm::allgatherv(const std::vector<uint8_t> &mine) const {
    std::vector<std::vector<uint8_t>> out(world);
    if (single()) {
        out[0] = mine;
        return out;
    }
    std::vector<uint64_t> sizes(world);
    const uint64_t my = mine.size();
    allgather(&my, sizes.data(), sizeof(uint64_t));
    uint64_t mx = 0;
    for (uint64_t s : sizes) mx = std::max(mx, s);
    if (mx == 0) return out;
    std::vector<uint8_t> send(mx, 0), recv(mx * world);
    if (my) memcpy(send.data(), mine.data(), my);
    allgather(send.data(), recv.data(), mx);
    for (uint32_t r = 0; r < world; r++) out[r].assign(recv.begin() + r * mx, recv.begin() + r * mx + sizes[r]);
    return out;
}

std::vector<uint8_t> ShardComm::bcast(uint32_t root, const std::vector<uint8_t> &mine) const {
    if (single()) return mine;
    auto all = allgatherv(rank == root ? mine : std::vector<uint8_t>());
    return std::move(all[root]);
}

void ShardComm::allgather_device(const void *d_mine, void *d_all, uint64_t bytes, ovm_stream_t stream) const {
    if (single()) throw OvmFail(1, "internal: device exchange with one rank");
    if (!abi->allgather_device) throw OvmFail(1, "ovm_shard_t: allgather_device is null");
    const int rc = abi->allgather_device(abi->ctx, d_mine, d_all, bytes, stream);
    if (rc != 0) throw OvmFail(rc, "shard exchange: allgather_device failed");
}

void ShardComm::alltoall_device(const void *d_send, void *d_recv, uint64_t bytes, ovm_stream_t stream) const {
    if (!has_alltoall()) throw OvmFail(1, "internal: alltoall_device is null");
    const int rc = abi->alltoall_device(abi->ctx, d_send, d_recv, bytes, stream);
    if (rc != 0) throw OvmFail(rc, "shard exchange: alltoall_device failed");
}

}  // namespace ovm

// ---------------------------------------------------------------------------------------------
// Shared-memory all-gather
// ---------------------------------------------------------------------------------------------
// Segment: one header line, then per rank two slots (sequence parity picks the slot). A rank that
// enters exchange k+2 has completed k+1, which needed every peer's contribution to k+1, which a peer
// only makes after it finished reading exchange k: so two slots per rank are enough and nobody ever
// overwrites data that is still being read.
namespace {
thread_local std::string g_shard_error;
}
extern "C" const char *ovm_hip_shard_last_error(void) { return g_shard_error.c_str(); }

struct ovm_shm_comm {
    struct alignas(64) Slot {
        uint64_t seq;
        uint64_t pad[7];
    };
    // The segment's first line. abort_code != 0: some rank gave up in the middle of a joint call (it threw, ran out of
    // memory, ...) and will not make its remaining contributions: every waiter returns at once instead of spinning
    // until the timeout, and the communicator is POISONED on every rank -- the sequence numbers of the ranks no longer
    // agree, so a later exchange could pair a fresh slot with a stale one and deliver bytes of the aborted proof.
    // Code and rank are ONE 64-bit word written by a single compare-and-swap (low half: code, high half: rank + 1), so a
    // peer that sees the code also sees who raised it.
    struct Header {
        uint64_t abort_word;
    };
    bool poisoned = false;
    Header *header() const { return reinterpret_cast<Header *>(base); }
    uint64_t abort_word() const { return __atomic_load_n(&header()->abort_word, __ATOMIC_ACQUIRE); }
    int fail(int code, const std::string &what) {
        poisoned = true;
        g_shard_error = "rank " + std::to_string(rank) + ": " + what;
        fprintf(stderr, "[ovm shard] %s\n", g_shard_error.c_str());
        return code;
    }
    int peer_aborted() {
        const uint64_t w = abort_word();
        return fail(OVM_ESHARD_PEER_ABORTED, "rank " + std::to_string((int)(w >> 32) - 1) + " aborted the joint call with code " +
                                                 std::to_string((uint32_t)w) + "; this communicator is unusable, create a new one");
    }
    int abort(int code) {
        uint64_t expected = 0;
        const uint64_t w = (uint64_t(rank + 1) << 32) | (code ? (uint32_t)code : 1u);
        __atomic_compare_exchange_n(&header()->abort_word, &expected, w, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);  // (the first abort wins)
        poisoned = true;
        return 0;
    }
    uint8_t *base = nullptr;
    size_t total = 0;
    uint32_t rank = 0, world = 1;
    uint64_t slot_bytes = 0;
    uint64_t seq = 0;
    std::string name;
    bool owns_mapping = false;
    // How long a rank waits for a peer that neither contributes nor aborts (a peer that FAILS raises the abort word and
    // is noticed within milliseconds, so this only bounds silent deaths and start-up skew: the first exchange of a job
    // waits for every rank's runtime, code objects and RCCL communicators -- tens of seconds on an 8-GPU node).
    double timeout_s = 300.0;

    size_t slot_stride() const { return sizeof(Slot) + ((slot_bytes + 63) & ~uint64_t(63)); }
    Slot *slot(uint32_t r, unsigned parity) const { return reinterpret_cast<Slot *>(base + 64 + (size_t)(2 * r + parity) * slot_stride()); }
    static size_t segment_bytes(uint32_t world, uint64_t slot_bytes) {
        const size_t stride = sizeof(Slot) + ((slot_bytes + 63) & ~uint64_t(63));
        return 64 + (size_t)2 * world * stride;
    }

    int exchange_chunk(const uint8_t *mine, uint8_t *all, uint64_t stride_all, uint64_t bytes) {
        if (poisoned) {
            g_shard_error = "rank " + std::to_string(rank) + ": the communicator was poisoned by an earlier failure (time-out or aborted peer); create a new one";
            return OVM_ESHARD_POISONED;
        }
        if (abort_word()) return peer_aborted();
        const uint64_t s = ++seq;
        const unsigned par = (unsigned)(s & 1);
        Slot *me = slot(rank, par);
        if (bytes) memcpy(reinterpret_cast<uint8_t *>(me) + sizeof(Slot), mine, bytes);
        __atomic_store_n(&me->seq, s, __ATOMIC_RELEASE);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t r = 0; r < world; r++) {
            Slot *sl = slot(r, par);
            uint64_t spins = 0;
            while (__atomic_load_n(&sl->seq, __ATOMIC_ACQUIRE) != s) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((++spins & 0x3FF) == 0) {
                    if (abort_word()) return peer_aborted();
                    if ((spins & 0x3FFF) == 0) {
                        sched_yield();
                        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        if (waited > timeout_s)
                            return fail(OVM_ESHARD_TIMEOUT, "timed out after " + std::to_string((int)waited) + " s waiting for rank " + std::to_string(r) +
                                                                " at exchange " + std::to_string(s) + " (OVM_SHARD_TIMEOUT_S)");
                    }
                }
            }
            if (bytes) memcpy(all + r * stride_all, reinterpret_cast<uint8_t *>(sl) + sizeof(Slot), bytes);
        }
        return 0;
    }
    int allgather(const void *mine, void *all, uint64_t bytes) {
        if (bytes == 0) return exchange_chunk(nullptr, nullptr, 0, 0);
        for (uint64_t off = 0; off < bytes; off += slot_bytes) {
            const uint64_t n = std::min(slot_bytes, bytes - off);
            const int rc = exchange_chunk((const uint8_t *)mine + off, (uint8_t *)all + off, bytes, n);
            if (rc) return rc;
        }
        return 0;
    }
};

extern "C" {

int ovm_hip_shm_comm_create(const char *name, uint32_t rank, uint32_t world, uint64_t slot_bytes, ovm_shm_comm_t **out) {
    if (!name || !out || world == 0 || rank >= world || slot_bytes == 0 || slot_bytes > (uint64_t(1) << 30)) return 1;
    const size_t total = ovm_shm_comm::segment_bytes(world, slot_bytes);
    // every rank opens with O_CREAT: a new segment is zero-filled, and sequence 0 means "nothing yet"
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 1;
    if (ftruncate(fd, (off_t)total) != 0) {
        close(fd);
        return 1;
    }
    void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;  // hipErrorOutOfMemory
    auto *c = new ovm_shm_comm();
    c->base = (uint8_t *)p;
    c->total = total;
    c->rank = rank;
    c->world = world;
    c->slot_bytes = slot_bytes;
    c->name = name;
    c->owns_mapping = true;
    if (const char *e = getenv("OVM_SHARD_TIMEOUT_S")) c->timeout_s = std::max(1.0, atof(e));  // (a peer that died: give up after this long)
    // handshake: once it completes every rank has mapped the segment, so the name can go away
    // (nothing is left behind in /dev/shm when a job dies later)
    const int rc = c->allgather(nullptr, nullptr, 0);
    if (rank == 0) shm_unlink(name);
    if (rc) {
        munmap(p, total);
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}

// The same communicator over memory the caller already shares between its ranks (threads of one
// process: `base` = zero-initialised buffer of ovm_hip_shm_segment_bytes(world, slot_bytes) bytes).
uint64_t ovm_hip_shm_segment_bytes(uint32_t world, uint64_t slot_bytes) { return ovm_shm_comm::segment_bytes(world, slot_bytes); }
int ovm_hip_shm_comm_attach(void *base, uint32_t rank, uint32_t world, uint64_t slot_bytes, ovm_shm_comm_t **out) {
    if (!base || !out || world == 0 || rank >= world || slot_bytes == 0) return 1;
    auto *c = new ovm_shm_comm();
    c->base = (uint8_t *)base;
    c->total = ovm_shm_comm::segment_bytes(world, slot_bytes);
    c->rank = rank;
    c->world = world;
    c->slot_bytes = slot_bytes;
    *out = c;
    return 0;
}

void ovm_hip_shm_comm_destroy(ovm_shm_comm_t *c) {
    if (!c) return;
    if (c->owns_mapping && c->base) munmap(c->base, c->total);
    delete c;
}

int ovm_hip_shm_comm_abort(void *comm, int code) {
    if (!comm) return 1;
    return static_cast<ovm_shm_comm *>(comm)->abort(code);
}
int ovm_hip_shm_comm_poisoned(const ovm_shm_comm_t *comm) { return comm && comm->poisoned ? 1 : 0; }

int ovm_hip_shm_allgather(void *comm, const void *mine, void *all, uint64_t bytes) {
    if (!comm || (bytes && (!mine || !all))) return 1;
    return static_cast<ovm_shm_comm *>(comm)->allgather(mine, all, bytes);
}

}  // extern "C"
