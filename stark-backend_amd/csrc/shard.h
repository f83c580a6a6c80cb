// Exchange layer of the sharded (one AIR per GPU) prover. Host only, no HIP.
//
// A joint proof over N ranks needs exactly two kinds of exchange:
//   * SMALL host data, hundreds of times per proof: the per-rank partial sums of every sumcheck
//     sub-round, roots, public values, column openings. Every rank runs the SAME Fiat-Shamir
//     transcript on the gathered values, so one all-gather per round is the whole protocol. The
//     reference has no peer for this (its only multi-instance pattern is independent thread replicas,
//     crates/cuda-backend/examples/keccakf.rs:88-131); what it fixes is that these values are summed
//     per AIR on the host (crates/cpu-backend/src/logup_zerocheck/mod.rs:1519-1557,1582-1597).
//     The ranks of one node share a POSIX shared-memory segment: a rank stores its contribution and a
//     sequence number, the others spin on it -- about a microsecond, against tens for a collective
//     library call per round.
//   * ONE bulk device exchange: the mu-batched WHIR vector (prover/whir.rs:95-132), all-gathered
//     with RCCL over xGMI by the host language through a callback (torch.distributed in this
//     repository, an RCCL binding in a Rust crate).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "openvm_hip.h"

namespace ovm {

// The communicator a sharded entry point works with: ovm_shard_t plus convenience wrappers.
// world == 1 (or a null ovm_shard_t) makes every call the identity.
struct ShardComm {
    uint32_t rank = 0, world = 1;
    const ovm_shard_t *abi = nullptr;

    ShardComm() {}
    explicit ShardComm(const ovm_shard_t *s) {
        if (s && s->world > 1) {
            abi = s;
            rank = s->rank;
            world = s->world;
        }
    }
    bool single() const { return world == 1; }

    // all[r * bytes .. (r+1) * bytes) = rank r's `mine`
    void allgather(const void *mine, void *all, uint64_t bytes) const;
    // rank-indexed byte strings of any length (a size exchange, then one padded exchange)
    std::vector<std::vector<uint8_t>> allgatherv(const std::vector<uint8_t> &mine) const;
    // every rank receives root's bytes
    std::vector<uint8_t> bcast(uint32_t root, const std::vector<uint8_t> &mine) const;
    // device buffers: d_all[r * bytes ..] = rank r's d_mine; stream-ordered after what was enqueued on `stream`
    void allgather_device(const void *d_mine, void *d_all, uint64_t bytes, ovm_stream_t stream) const;
    // optional: d_recv[s * bytes ..] = rank s's d_send[rank * bytes ..]
    bool has_alltoall() const { return abi && abi->alltoall_device; }
    void alltoall_device(const void *d_send, void *d_recv, uint64_t bytes, ovm_stream_t stream) const;
};

// Append / read plain values in a byte string (the payload format of the small exchanges).
struct Packer {
    std::vector<uint8_t> buf;
    template <class T>
    void put(const T &v) {
        const size_t o = buf.size();
        buf.resize(o + sizeof(T));
        memcpy(buf.data() + o, &v, sizeof(T));
    }
    template <class T>
    void put_vec(const std::vector<T> &v) {
        put<uint64_t>(v.size());
        const size_t o = buf.size();
        buf.resize(o + v.size() * sizeof(T));
        if (!v.empty()) memcpy(buf.data() + o, v.data(), v.size() * sizeof(T));
    }
};
struct Unpacker {
    std::vector<uint8_t> owned;  // (a copy: messages are small, and a temporary can be passed safely)
    const uint8_t *p;
    size_t n, at = 0;
    explicit Unpacker(std::vector<uint8_t> b) : owned(std::move(b)), p(owned.data()), n(owned.size()) {}
    Unpacker(const Unpacker &) = delete;
    Unpacker &operator=(const Unpacker &) = delete;
    bool done() const { return at >= n; }
    void need(size_t k) const;
    template <class T>
    T get() {
        need(sizeof(T));
        T v;
        memcpy(&v, p + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    template <class T>
    std::vector<T> get_vec() {
        const uint64_t k = get<uint64_t>();
        need(k * sizeof(T));
        std::vector<T> v(k);
        if (k) memcpy((void *)v.data(), p + at, k * sizeof(T));
        at += k * sizeof(T);
        return v;
    }
};

}  // namespace ovm
