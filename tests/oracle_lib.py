"""ctypes loader for the CPU oracle (oracle/_build/liboracle.so). Test infrastructure only."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_int, c_uint32, c_uint64, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
P = 2013265921

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR)
            if f.endswith((".hpp", ".cpp", ".inc"))]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        build()
    lib = ctypes.CDLL(LIB)
    lib.orc_to_mont.restype = c_uint32
    lib.orc_from_mont.restype = c_uint32
    lib.orc_two_adic_generator.restype = c_uint32
    lib.orc_stacked_matrix.restype = c_uint64
    lib.orc_transcript_new.restype = c_void_p
    lib.orc_transcript_sample.restype = c_uint32
    lib.orc_transcript_grind.restype = c_uint32
    _lib = Oracle(lib)
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def permute(self, state):
        s = np.ascontiguousarray(state, dtype=np.uint32).copy()
        self.lib.orc_poseidon2_permute(_p(s))
        return s

    def permute_rc(self, state, initial, terminal, internal):
        """The oracle's layer code over caller-supplied (canonical) round constants."""
        s = np.ascontiguousarray(state, dtype=np.uint32).copy()
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint32) for x in (initial, terminal, internal))
        assert a.size == 64 and b.size == 64 and c.size == 13
        self.lib.orc_poseidon2_permute_rc(_p(s), _p(a), _p(b), _p(c))
        return s

    def hash_slice(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.uint32)
        out = np.zeros(8, dtype=np.uint32)
        self.lib.orc_hash_slice(_p(v), c_uint64(v.size), _p(out))
        return out

    def compress(self, l, r):
        out = np.zeros(8, dtype=np.uint32)
        self.lib.orc_compress(_p(np.ascontiguousarray(l, dtype=np.uint32)),
                              _p(np.ascontiguousarray(r, dtype=np.uint32)), _p(out))
        return out

    def dft(self, a, inverse=False):
        v = np.ascontiguousarray(a, dtype=np.uint32).copy()
        self.lib.orc_dft(_p(v), c_uint64(v.size), c_int(1 if inverse else 0))
        return v

    def ef_mul(self, a, b):
        out = np.zeros(4, dtype=np.uint32)
        self.lib.orc_ef_mul(_p(np.ascontiguousarray(a, dtype=np.uint32)),
                            _p(np.ascontiguousarray(b, dtype=np.uint32)), _p(out))
        return out

    def ef_inv(self, a):
        out = np.zeros(4, dtype=np.uint32)
        self.lib.orc_ef_inv(_p(np.ascontiguousarray(a, dtype=np.uint32)), _p(out))
        return out

    def stacked_matrix(self, l_skip, n_stack, traces):
        """traces: list of (col-major flat u32 array, width, height). Returns (stacked flat,
        width, layout rows[(mat, col, scol, srow, log_h)])."""
        n = len(traces)
        arrs = [np.ascontiguousarray(t[0], dtype=np.uint32) for t in traces]
        ptrs = (c_void_p * n)(*[a.ctypes.data for a in arrs])
        widths = np.array([t[1] for t in traces], dtype=np.uint64)
        heights = np.array([t[2] for t in traces], dtype=np.uint64)
        ncols = int(sum(t[1] for t in traces))
        layout = np.zeros(5 * ncols, dtype=np.uint64)
        w = self.lib.orc_stacked_matrix(c_uint32(l_skip), c_uint32(n_stack), c_uint32(n), ptrs,
                                        _p(widths), _p(heights), None, _p(layout))
        out = np.zeros(int(w) << (l_skip + n_stack), dtype=np.uint32)
        self.lib.orc_stacked_matrix(c_uint32(l_skip), c_uint32(n_stack), c_uint32(n), ptrs,
                                    _p(widths), _p(heights), _p(out), None)
        return out, int(w), layout.reshape(-1, 5)

    def rs_code_matrix(self, l_skip, log_blowup, evals, width, height):
        e = np.ascontiguousarray(evals, dtype=np.uint32)
        out = np.zeros(width * (height << log_blowup), dtype=np.uint32)
        self.lib.orc_rs_code_matrix(c_uint32(l_skip), c_uint32(log_blowup), _p(e), c_uint64(width),
                                    c_uint64(height), _p(out))
        return out

    def merkle_tree(self, matrix, width, height, rows_per_query, ext=False):
        m = np.ascontiguousarray(matrix, dtype=np.uint32)
        qs = height // rows_per_query
        out = np.zeros((2 * qs - 1) * 8, dtype=np.uint32)
        fn = self.lib.orc_merkle_tree_ext if ext else self.lib.orc_merkle_tree
        fn(_p(m), c_uint64(width), c_uint64(height), c_uint64(rows_per_query), _p(out))
        layers = []
        off = 0
        n = qs
        while n >= 1:
            layers.append(out[off * 8:(off + n) * 8].reshape(n, 8))
            off += n
            n //= 2
        return layers

    def transcript(self):
        return OracleTranscript(self.lib)


class OracleTranscript:
    def __init__(self, lib):
        self.lib = lib
        self.h = c_void_p(lib.orc_transcript_new())

    def __del__(self):
        try:
            self.lib.orc_transcript_free(self.h)
        except Exception:
            pass

    def observe(self, x):
        self.lib.orc_transcript_observe(self.h, c_uint32(int(x)))

    def sample(self):
        return int(self.lib.orc_transcript_sample(self.h))

    def grind(self, bits):
        return int(self.lib.orc_transcript_grind(self.h, c_uint32(bits)))

    def state(self):
        st = np.zeros(16, dtype=np.uint32)
        a = c_uint32(0)
        s = c_uint32(0)
        self.lib.orc_transcript_state(self.h, _p(st), ctypes.byref(a), ctypes.byref(s))
        return st, a.value, s.value


def _prove(self, blob, traces, proof_cap=1 << 26):
    """traces: list of (air_idx, canonical col-major flat u32 trace, height, public values list
    [, cached partitions concatenated, col-major]). Returns dict(proof=bytes, root, r, u, pow)."""
    lib = self.lib
    n = len(traces)
    blob = np.ascontiguousarray(blob, dtype=np.uint32)
    arrs = [np.ascontiguousarray(t[1], dtype=np.uint32) for t in traces]
    # small cases: a handful of threads (a 256-thread team per tiny loop costs more than the loop)
    cells = sum(a.size for a in arrs) + sum(np.size(t[4]) for t in traces if len(t) > 4 and t[4] is not None)
    lib.orc_set_num_threads(c_int(int(os.environ.get("ORC_THREADS", "0")) or (8 if cells < (1 << 22) else 0)))
    pvs = [np.ascontiguousarray(t[3], dtype=np.uint32) if len(t[3]) else np.zeros(1, np.uint32) for t in traces]
    air_idx = np.array([t[0] for t in traces], dtype=np.uint32)
    heights = np.array([t[2] for t in traces], dtype=np.uint64)
    npv = np.array([len(t[3]) for t in traces], dtype=np.uint32)
    tptr = (c_void_p * n)(*[a.ctypes.data for a in arrs])
    pptr = (c_void_p * n)(*[a.ctypes.data for a in pvs])
    carrs = [np.ascontiguousarray(t[4], dtype=np.uint32) if len(t) > 4 and t[4] is not None else None for t in traces]
    cptr = (c_void_p * n)(*[a.ctypes.data if a is not None else None for a in carrs])
    out = np.zeros(proof_cap, dtype=np.uint8)
    plen = c_uint64(0)
    root = np.zeros(8, np.uint32)
    r = np.zeros(4 * 64, np.uint32); u = np.zeros(4 * 64, np.uint32); pw = np.zeros(256, np.uint32)
    rl = c_uint32(0); ul = c_uint32(0); pl = c_uint32(0)
    lib.orc_last_error.restype = ctypes.c_char_p
    try:
        rc = lib.orc_prove(_p(blob), c_uint64(blob.size), c_uint32(n), _p(air_idx), tptr, _p(heights), cptr, pptr,
                           _p(npv), _p(out), c_uint64(proof_cap), ctypes.byref(plen), _p(root), _p(r),
                           ctypes.byref(rl), _p(u), ctypes.byref(ul), _p(pw), ctypes.byref(pl))
    finally:
        lib.orc_set_num_threads(c_int(0))  # the other entry points (commit, verify) keep the runtime's default
    if rc != 0:
        raise RuntimeError("orc_prove: " + lib.orc_last_error().decode())
    return dict(proof=out[:plen.value].tobytes(), root=root, r=r[:4 * rl.value].reshape(-1, 4),
                u=u[:4 * ul.value].reshape(-1, 4), pow=pw[:pl.value].copy())


def _verify(self, blob, proof: bytes):
    lib = self.lib
    blob = np.ascontiguousarray(blob, dtype=np.uint32)
    buf = np.frombuffer(proof, dtype=np.uint8)
    lib.orc_last_error.restype = ctypes.c_char_p
    rc = lib.orc_verify(_p(blob), c_uint64(blob.size), _p(buf), c_uint64(buf.size))
    if rc != 0:
        raise RuntimeError("orc_verify: " + lib.orc_last_error().decode())


Oracle.prove = _prove
Oracle.verify = _verify
