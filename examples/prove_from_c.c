/* A plain-C caller of libopenvm_hip.so: no PyTorch, no HIP headers, nothing but include/openvm_hip.h
 * -- the way the Rust crate binds it. Reads a proving-key blob and one canonical
 * column-major trace from files, uploads the trace as Montgomery words, proves, writes the proof.
 *
 *   gcc -O2 -I include examples/prove_from_c.c -o examples/prove_from_c -L stark-backend_amd -lopenvm_hip \
 *       -Wl,-rpath,'$ORIGIN/../stark-backend_amd'
 *   examples/prove_from_c key.u32 trace.u32 <air_idx> <height> proof.bin
 *
 * tests/test_gpu_c_example.py builds and runs it and compares the proof with ovm_hip_prove called
 * through the Python binding. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "openvm_hip.h"

#define BB_P 2013265921u

static uint32_t *read_u32(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint32_t *buf = (uint32_t *)malloc((size_t)bytes);
    if (fread(buf, 1, (size_t)bytes, f) != (size_t)bytes) { perror("read"); exit(2); }
    fclose(f);
    *n = (size_t)bytes / 4;
    return buf;
}

#define CHECK(call)                                                                                       \
    do {                                                                                                  \
        int rc_ = (call);                                                                                 \
        if (rc_ != 0) {                                                                                   \
            fprintf(stderr, "%s failed: %d (%s) [%s]\n", #call, rc_, ovm_hip_error_name(rc_), ovm_hip_last_error()); \
            exit(1);                                                                                      \
        }                                                                                                 \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 6) {
        fprintf(stderr, "usage: %s key.u32 trace.u32 air_idx height proof.bin\n", argv[0]);
        return 2;
    }
    size_t key_words, trace_words;
    uint32_t *key = read_u32(argv[1], &key_words);
    uint32_t *trace = read_u32(argv[2], &trace_words);
    const uint32_t air_idx = (uint32_t)atoi(argv[3]);
    const uint64_t height = strtoull(argv[4], NULL, 10);

    uint32_t cus = 0;
    CHECK(ovm_hip_init(0, &cus));
    fprintf(stderr, "ABI %u, %u compute units\n", ovm_hip_abi_version(), cus);

    /* canonical -> Montgomery (x * 2^32 mod p): what a Rust `BabyBear` holds in memory */
    for (size_t i = 0; i < trace_words; i++) trace[i] = (uint32_t)((((uint64_t)trace[i] % BB_P) << 32) % BB_P);

    ovm_stream_t stream;
    CHECK(ovm_hip_stream_create(&stream));
    void *d_trace = NULL;
    CHECK(ovm_hip_malloc(&d_trace, trace_words * 4));
    CHECK(ovm_hip_memcpy_h2d(d_trace, trace, trace_words * 4, stream));

    ovm_pk_t *pk = NULL;
    CHECK(ovm_hip_pk_load(key, key_words, &pk, stream));

    ovm_trace_t t;
    t.air_idx = air_idx;
    t.num_public_values = 0;
    t.d_common_main = (const uint32_t *)d_trace;
    t.height = height;
    t.public_values = NULL;
    t.num_cached = 0;
    t._pad = 0;
    t.cached = NULL;

    const uint64_t cap = 1u << 24;
    uint8_t *proof = (uint8_t *)malloc(cap);
    uint64_t len = 0, sections[5];
    CHECK(ovm_hip_prove(pk, 1, &t, NULL, proof, cap, &len, sections, stream));
    fprintf(stderr, "proof: %llu bytes (sections end at %llu %llu %llu %llu %llu)\n", (unsigned long long)len,
            (unsigned long long)sections[0], (unsigned long long)sections[1], (unsigned long long)sections[2],
            (unsigned long long)sections[3], (unsigned long long)sections[4]);
    FILE *f = fopen(argv[5], "wb");
    if (!f || fwrite(proof, 1, len, f) != len) { perror(argv[5]); return 2; }
    fclose(f);

    ovm_hip_pk_free(pk);
    CHECK(ovm_hip_free(d_trace));
    CHECK(ovm_hip_stream_destroy(stream));
    free(proof);
    free(trace);
    free(key);
    return 0;
}
