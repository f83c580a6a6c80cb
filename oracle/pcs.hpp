// ORACLE — test infrastructure only (see oracle/README.md).
//
// Stacked polynomial commitment: layout, stacking, Reed-Solomon encode and Merkle tree, restated
// from the column-major SPEC implementation:
//   crates/stark-backend/src/prover/stacked_pcs.rs:18-75     StackedLayout / StackedSlice
//   crates/stark-backend/src/prover/stacked_pcs.rs:144-203   StackedLayout::new
//   crates/stark-backend/src/prover/stacked_pcs.rs:294-335   stacked_matrix
//   crates/stark-backend/src/prover/stacked_pcs.rs:341-367   rs_code_matrix
//   crates/stark-backend/src/prover/stacked_pcs.rs:369-405   root / query_merkle_proof
//   crates/stark-backend/src/prover/stacked_pcs.rs:413-485   MerkleTree::new (strided first layers)
//   crates/stark-backend/src/prover/stacked_pcs.rs:516-540   get_opened_rows
#pragma once
#include <omp.h>
#include <stdexcept>

#include "poly.hpp"
#include "poseidon2.hpp"

namespace oracle {

template <class T>
struct ColMat {
    std::vector<T> values;  // values[col * height + row]
    size_t width = 0, height = 0;
    ColMat() {}
    ColMat(std::vector<T> v, size_t w) : values(std::move(v)), width(w) {
        height = w ? values.size() / w : 0;
        assert(w == 0 || values.size() % w == 0);
    }
    const T *column(size_t j) const { return values.data() + j * height; }
    T *column(size_t j) { return values.data() + j * height; }
    const T &at(size_t r, size_t c) const { return values[c * height + r]; }
};

struct StackedSlice {
    size_t col_idx = 0, row_idx = 0;
    unsigned log_height = 0;
    static size_t len_of(unsigned log_height, unsigned l_skip) {
        return size_t(1) << (l_skip <= log_height ? log_height : l_skip);
    }
    size_t len(unsigned l_skip) const { return len_of(log_height, l_skip); }
    size_t stride(unsigned l_skip) const {
        return size_t(1) << (l_skip > log_height ? l_skip - log_height : 0);
    }
};

struct SortedCol {
    size_t mat_idx, col_idx;
    StackedSlice slice;
};

struct StackedLayout {
    unsigned l_skip = 0;
    size_t height = 0, width = 0;
    std::vector<SortedCol> sorted_cols;
    std::vector<size_t> mat_starts;

    StackedLayout() {}
    // stacked_pcs.rs:144-203 ; `sorted` = (width, log_height), descending log_height
    StackedLayout(unsigned l_skip_, unsigned log_stacked_height,
                  const std::vector<std::pair<size_t, unsigned>> &sorted)
        : l_skip(l_skip_) {
        size_t col_idx = 0, row_idx = 0;
        size_t H = size_t(1) << log_stacked_height;
        for (size_t mat_idx = 0; mat_idx < sorted.size(); mat_idx++) {
            auto [w, log_ht] = sorted[mat_idx];
            mat_starts.push_back(sorted_cols.size());
            if (w == 0) continue;
            if (log_ht > log_stacked_height) throw std::runtime_error("LayoutHeightExceeded");
            for (size_t j = 0; j < w; j++) {
                size_t slice_len = StackedSlice::len_of(log_ht, l_skip);
                if (row_idx + slice_len > H) {
                    if (row_idx != H) throw std::runtime_error("LayoutRowOverflow");
                    col_idx++;
                    row_idx = 0;
                }
                StackedSlice s;
                s.col_idx = col_idx;
                s.row_idx = row_idx;
                s.log_height = log_ht;
                sorted_cols.push_back({mat_idx, j, s});
                row_idx += slice_len;
            }
        }
        height = H;
        width = col_idx + (row_idx != 0 ? 1 : 0);
    }
    const StackedSlice *get(size_t mat_idx, size_t col_idx) const {
        size_t idx = mat_starts[mat_idx];
        if (idx + col_idx >= sorted_cols.size()) return nullptr;
        return &sorted_cols[idx + col_idx].slice;
    }
};

// stacked_pcs.rs:294-335
inline std::pair<ColMat<Fp>, StackedLayout> stacked_matrix(
    unsigned l_skip, unsigned n_stack, const std::vector<const ColMat<Fp> *> &traces) {
    std::vector<std::pair<size_t, unsigned>> meta;
    for (auto *t : traces) meta.push_back({t->width, log2_strict(t->height)});
    StackedLayout layout(l_skip, l_skip + n_stack, meta);
    size_t total_cells = 0;
    for (auto *t : traces)
        total_cells += std::max(t->height, size_t(1) << l_skip) * t->width;
    size_t height = size_t(1) << (l_skip + n_stack);
    size_t width = (total_cells + height - 1) / height;
    std::vector<Fp> q(width * height);
    for (auto &sc : layout.sorted_cols) {
        const StackedSlice &s = sc.slice;
        size_t start = s.col_idx * height + s.row_idx;
        const Fp *t_col = traces[sc.mat_idx]->column(sc.col_idx);
        size_t t_len = size_t(1) << s.log_height;
        if (s.log_height >= l_skip) {
            std::copy(t_col, t_col + t_len, q.begin() + start);
        } else {
            size_t stride = s.stride(l_skip);
            for (size_t i = 0; i < t_len; i++) q[start + i * stride] = t_col[i];
        }
    }
    return {ColMat<Fp>(std::move(q), width), layout};
}

// stacked_pcs.rs:341-367 : natural-order codeword over the subgroup of size height << log_blowup
inline ColMat<Fp> rs_code_matrix(unsigned l_skip, unsigned log_blowup, const ColMat<Fp> &evals) {
    size_t height = evals.height, rs_height = height << log_blowup;
    std::vector<Fp> out(rs_height * evals.width);
    // one thread per column -- except for a handful of tall columns (the 1-column common main of a partitioned AIR), whose
    // transforms spread over the cores themselves (poly.hpp: dft_generic). Measured on the 256-core host at config D
    // (25 columns of 2^25 points): by column 26 s, column after column with 256-thread transforms 55 s.
    const bool by_column = evals.width >= 4 || rs_height < (size_t(1) << 16);
#pragma omp parallel for schedule(dynamic) if (by_column)
    for (size_t j = 0; j < evals.width; j++) {
        std::vector<Fp> coeffs = eval_to_coeff_rs_message(l_skip, evals.column(j), height);
        coeffs.resize(rs_height, Fp::zero());
        coeffs = dft(std::move(coeffs));
        std::copy(coeffs.begin(), coeffs.end(), out.begin() + j * rs_height);
    }
    return ColMat<Fp>(std::move(out), evals.width);
}

// Merkle tree over rows of a column-major matrix whose entries are F or EF (EF rows are
// flattened coefficient-wise, stacked_pcs.rs:435-443).
template <class T>
struct MerkleTree {
    ColMat<T> backing;
    std::vector<std::vector<Digest>> digest_layers;
    size_t rows_per_query = 1;

    static void push_coeffs(std::vector<Fp> &v, const Fp &x) { v.push_back(x); }
    static void push_coeffs(std::vector<Fp> &v, const EF &x) {
        for (int i = 0; i < 4; i++) v.push_back(x.c[i]);
    }

    MerkleTree() {}
    MerkleTree(ColMat<T> matrix, size_t rpq) : backing(std::move(matrix)), rows_per_query(rpq) {
        size_t height = backing.height;
        if (height == 0) throw std::runtime_error("MerkleTreeEmptyMatrix");
        if (rpq & (rpq - 1)) throw std::runtime_error("MerkleTreeRowsPerQueryNotPow2");
        size_t num_leaves = size_t(1) << log2_ceil(height);
        if (rpq > num_leaves) throw std::runtime_error("MerkleTreeRowsPerQueryExceeded");
        std::vector<Digest> row_hashes(num_leaves);
#pragma omp parallel
        {
            std::vector<Fp> in;  // (one buffer per thread: an allocation per row is 2^25 trips through malloc on 256 threads)
#pragma omp for schedule(static)
            for (size_t r = 0; r < num_leaves; r++) {
                in.clear();
                for (size_t c = 0; c < backing.width; c++)
                    push_coeffs(in, r < height ? backing.at(r, c) : T());
                row_hashes[r] = hash_slice(in);
            }
        }
        size_t query_stride = num_leaves / rpq;
        std::vector<Digest> layer = std::move(row_hashes);
        for (unsigned l = 0; l < log2_strict(rpq); l++) {
            std::vector<Digest> next(layer.size() / 2);
#pragma omp parallel for
            for (size_t i = 0; i < next.size(); i++) {
                size_t x = i / query_stride, y = i % query_stride;
                next[i] = compress(layer[2 * x * query_stride + y],
                                   layer[(2 * x + 1) * query_stride + y]);
            }
            layer = std::move(next);
        }
        digest_layers.push_back(std::move(layer));
        while (digest_layers.back().size() > 1) {
            const auto &prev = digest_layers.back();
            std::vector<Digest> next(prev.size() / 2);
#pragma omp parallel for
            for (size_t i = 0; i < next.size(); i++) next[i] = compress(prev[2 * i], prev[2 * i + 1]);
            digest_layers.push_back(std::move(next));
        }
    }
    size_t query_stride() const { return digest_layers[0].size(); }
    size_t proof_depth() const { return digest_layers.size() - 1; }
    Digest root() const { return digest_layers.back()[0]; }
    std::vector<Digest> query_merkle_proof(size_t query_idx) const {
        if (query_idx >= query_stride()) throw std::runtime_error("MerkleTreeQueryOutOfBounds");
        std::vector<Digest> proof;
        size_t idx = query_idx;
        for (size_t l = 0; l < proof_depth(); l++) {
            proof.push_back(digest_layers[l][idx ^ 1]);
            idx >>= 1;
        }
        return proof;
    }
    // rows { index + t * query_stride }, t in 0..rows_per_query
    std::vector<std::vector<T>> get_opened_rows(size_t index) const {
        size_t qs = query_stride();
        if (index >= qs) throw std::runtime_error("MerkleTreeOpenedRowsOutOfBounds");
        std::vector<std::vector<T>> rows;
        for (size_t t = 0; t < rows_per_query; t++) {
            size_t r = t * qs + index;
            std::vector<T> row(backing.width);
            for (size_t c = 0; c < backing.width; c++)
                row[c] = r < backing.height ? backing.at(r, c) : T();
            rows.push_back(std::move(row));
        }
        return rows;
    }
};

struct StackedPcsData {
    StackedLayout layout;
    ColMat<Fp> matrix;  // stacked evaluations, height 2^(l_skip+n_stack)
    MerkleTree<Fp> tree;
    Digest commit() const { return tree.root(); }
};

// stacked_pcs.rs:115-134
inline StackedPcsData stacked_commit(unsigned l_skip, unsigned n_stack, unsigned log_blowup,
                                     unsigned k_whir,
                                     const std::vector<const ColMat<Fp> *> &traces) {
    auto [q, layout] = stacked_matrix(l_skip, n_stack, traces);
    ColMat<Fp> rs = rs_code_matrix(l_skip, log_blowup, q);
    StackedPcsData d;
    d.layout = layout;
    d.matrix = std::move(q);
    d.tree = MerkleTree<Fp>(std::move(rs), size_t(1) << k_whir);
    return d;
}

}  // namespace oracle
