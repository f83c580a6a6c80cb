// ORACLE — test infrastructure only (see oracle/README.md).
// extern "C" surface for tests/ (ctypes). All field values cross this API in CANONICAL form
// (0 <= x < P); tests convert to/from the Montgomery words the product ABI uses.
#include <cstring>

#include "pcs.hpp"
#include "prover.hpp"
#include "verifier.hpp"

using namespace oracle;

static std::vector<Fp> to_fp(const uint32_t *p, size_t n) {
    std::vector<Fp> v(n);
    for (size_t i = 0; i < n; i++) v[i] = Fp((uint64_t)p[i]);
    return v;
}

#if defined(_OPENMP)
#include <omp.h>
#endif

extern "C" {

// Threads of the following calls (0 = the OpenMP runtime's own default). Small test cases on a many-core host spend their time
// starting and joining 256-thread teams thousands of times; tests/oracle_lib.py picks the count by problem size.
void orc_set_num_threads(int n) {
#if defined(_OPENMP)
    static const int initial = omp_get_max_threads();  // what the runtime chose before anybody asked (0 restores it)
    omp_set_num_threads(n > 0 ? (n < initial ? n : initial) : initial);
#else
    (void)n;
#endif
}


uint32_t orc_modulus() { return P; }
uint32_t orc_to_mont(uint32_t x) { return Fp((uint64_t)x).to_mont(); }
uint32_t orc_from_mont(uint32_t x) { return Fp::from_mont(x).v; }
uint32_t orc_two_adic_generator(uint32_t bits) { return two_adic_generator(bits).v; }

void orc_ef_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    EF x(Fp((uint64_t)a[0]), Fp((uint64_t)a[1]), Fp((uint64_t)a[2]), Fp((uint64_t)a[3]));
    EF y(Fp((uint64_t)b[0]), Fp((uint64_t)b[1]), Fp((uint64_t)b[2]), Fp((uint64_t)b[3]));
    EF z = x * y;
    for (int i = 0; i < 4; i++) out[i] = z.c[i].v;
}
void orc_ef_inv(const uint32_t *a, uint32_t *out) {
    EF x(Fp((uint64_t)a[0]), Fp((uint64_t)a[1]), Fp((uint64_t)a[2]), Fp((uint64_t)a[3]));
    EF z = x.inv();
    for (int i = 0; i < 4; i++) out[i] = z.c[i].v;
}

void orc_poseidon2_permute(uint32_t *state16) {
    Fp s[16];
    for (int i = 0; i < 16; i++) s[i] = Fp((uint64_t)state16[i]);
    poseidon2_permute(s);
    for (int i = 0; i < 16; i++) state16[i] = s[i].v;
}
void orc_poseidon2_permute_rc(uint32_t *state16, const uint32_t *initial64, const uint32_t *terminal64,
                              const uint32_t *internal13) {
    Fp s[16];
    for (int i = 0; i < 16; i++) s[i] = Fp((uint64_t)state16[i]);
    poseidon2_permute_rc(s, initial64, terminal64, internal13);
    for (int i = 0; i < 16; i++) state16[i] = s[i].v;
}
void orc_poseidon2_constants(uint32_t *initial64, uint32_t *terminal64, uint32_t *internal13,
                              uint32_t *diag16) {
    memcpy(initial64, RC_INITIAL, sizeof(RC_INITIAL));
    memcpy(terminal64, RC_TERMINAL, sizeof(RC_TERMINAL));
    memcpy(internal13, RC_INTERNAL, sizeof(RC_INTERNAL));
    memcpy(diag16, INTERNAL_DIAG, sizeof(INTERNAL_DIAG));
}
void orc_hash_slice(const uint32_t *vals, uint64_t n, uint32_t *out8) {
    auto v = to_fp(vals, n);
    Digest d = hash_slice(v);
    for (int i = 0; i < 8; i++) out8[i] = d[i].v;
}
void orc_compress(const uint32_t *l, const uint32_t *r, uint32_t *out8) {
    Digest a, b;
    for (int i = 0; i < 8; i++) {
        a[i] = Fp((uint64_t)l[i]);
        b[i] = Fp((uint64_t)r[i]);
    }
    Digest d = compress(a, b);
    for (int i = 0; i < 8; i++) out8[i] = d[i].v;
}

// natural-order (i)DFT of `n` base-field values
void orc_dft(uint32_t *a, uint64_t n, int inverse) {
    auto v = dft_generic(to_fp(a, n), inverse != 0);
    for (size_t i = 0; i < n; i++) a[i] = v[i].v;
}

// Stacking. traces[i] is column-major widths[i] x heights[i]. Returns stacked width; `out` (if
// non-null) must hold width * 2^(l_skip+n_stack) words. `layout_out` (if non-null) receives
// 5 words per sorted column: mat_idx, col_idx, stacked col, stacked row, log_height.
uint64_t orc_stacked_matrix(uint32_t l_skip, uint32_t n_stack, uint32_t n_traces,
                            const uint32_t *const *traces, const uint64_t *widths,
                            const uint64_t *heights, uint32_t *out, uint64_t *layout_out) {
    std::vector<ColMat<Fp>> mats;
    for (uint32_t i = 0; i < n_traces; i++)
        mats.emplace_back(to_fp(traces[i], widths[i] * heights[i]), widths[i]);
    std::vector<const ColMat<Fp> *> refs;
    for (auto &m : mats) refs.push_back(&m);
    auto [q, layout] = stacked_matrix(l_skip, n_stack, refs);
    if (out)
        for (size_t i = 0; i < q.values.size(); i++) out[i] = q.values[i].v;
    if (layout_out) {
        size_t k = 0;
        for (auto &sc : layout.sorted_cols) {
            layout_out[k++] = sc.mat_idx;
            layout_out[k++] = sc.col_idx;
            layout_out[k++] = sc.slice.col_idx;
            layout_out[k++] = sc.slice.row_idx;
            layout_out[k++] = sc.slice.log_height;
        }
    }
    return q.width;
}

// Column-major evals (width x height) -> column-major natural-order codeword (width x height<<lb)
void orc_rs_code_matrix(uint32_t l_skip, uint32_t log_blowup, const uint32_t *evals, uint64_t width,
                        uint64_t height, uint32_t *out) {
    ColMat<Fp> m(to_fp(evals, width * height), width);
    ColMat<Fp> rs = rs_code_matrix(l_skip, log_blowup, m);
    for (size_t i = 0; i < rs.values.size(); i++) out[i] = rs.values[i].v;
}

// Merkle tree of a column-major base-field matrix. `layers_out` receives all stored digest layers
// concatenated (query_stride + query_stride/2 + ... + 1 digests, 8 words each).
void orc_merkle_tree(const uint32_t *matrix, uint64_t width, uint64_t height,
                     uint64_t rows_per_query, uint32_t *layers_out) {
    MerkleTree<Fp> t(ColMat<Fp>(to_fp(matrix, width * height), width), rows_per_query);
    size_t k = 0;
    for (auto &layer : t.digest_layers)
        for (auto &d : layer)
            for (int i = 0; i < 8; i++) layers_out[k++] = d[i].v;
}
// Same for an extension-field matrix (4 words per entry).
void orc_merkle_tree_ext(const uint32_t *matrix, uint64_t width, uint64_t height,
                         uint64_t rows_per_query, uint32_t *layers_out) {
    std::vector<EF> v(width * height);
    for (size_t i = 0; i < v.size(); i++)
        for (int j = 0; j < 4; j++) v[i].c[j] = Fp((uint64_t)matrix[4 * i + j]);
    MerkleTree<EF> t(ColMat<EF>(std::move(v), width), rows_per_query);
    size_t k = 0;
    for (auto &layer : t.digest_layers)
        for (auto &d : layer)
            for (int i = 0; i < 8; i++) layers_out[k++] = d[i].v;
}

// --- transcript ---------------------------------------------------------------------------
void *orc_transcript_new() { return new Transcript(); }
void orc_transcript_free(void *t) { delete (Transcript *)t; }
void orc_transcript_observe(void *t, uint32_t x) { ((Transcript *)t)->observe(Fp((uint64_t)x)); }
uint32_t orc_transcript_sample(void *t) { return ((Transcript *)t)->sample().v; }
uint32_t orc_transcript_grind(void *t, uint32_t bits) { return ((Transcript *)t)->grind(bits).v; }
void orc_transcript_state(void *t, uint32_t *state16, uint32_t *absorb_idx, uint32_t *sample_idx) {
    Transcript *ts = (Transcript *)t;
    for (int i = 0; i < 16; i++) state16[i] = ts->state[i].v;
    *absorb_idx = ts->absorb_idx;
    *sample_idx = ts->sample_idx;
}

}  // extern "C"

#include "capi_prover.inc"
