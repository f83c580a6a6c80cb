"""MI355X (gfx950) prover backend for openvm-org/stark-backend v2 -- Python host side.

This package is a thin ctypes binding over the C ABI declared in ``include/openvm_hip.h``
(``libopenvm_hip.so``, built in-tree by ``build.sh``). It mirrors the reference's backend
surface for the prove hot path (``TraceCommitter::commit``, ``MultiRapProver::prove_rap_constraints``,
``OpeningProver::prove_openings`` -- reference: crates/stark-backend/src/prover/hal.rs:84-138) and is
what the tests and ``bench.py`` drive. PyTorch is used only as the owner of device memory and
streams.

The library is REQUIRED: importing the compute entry points without it raises. There is no CPU
fallback in this package (the CPU restatement under ``oracle/`` is test infrastructure).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_int, c_size_t, c_uint32, c_uint64,
                    c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OVM_HIP_LIB") or os.path.join(_HERE, "libopenvm_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

P = 2013265921  # 15 * 2^27 + 1
MONTY_BITS = 32


class OvmError(RuntimeError):
    """A non-zero return code from the C ABI (a hipError_t value)."""

    def __init__(self, code: int, what: str):
        self.code = code
        name = _lib().ovm_hip_error_name(code) if _LIB is not None else b"?"
        super().__init__(f"{what} failed: {code} ({name.decode() if name else '?'})")


class TraceStruct(Structure):
    """``ovm_trace_t`` (reference: AirProvingContext, prover/types.rs:59-73)."""

    _fields_ = [("air_idx", c_uint32), ("num_public_values", c_uint32), ("d_common_main", c_void_p),
                ("height", c_uint64), ("public_values", c_void_p), ("num_cached", c_uint32), ("_pad", c_uint32),
                ("cached", c_void_p)]


class PcsParams(Structure):
    """``ovm_pcs_params_t``: the parameters a commitment depends on (config.rs:51-65)."""

    _fields_ = [("l_skip", c_uint32), ("n_stack", c_uint32), ("log_blowup", c_uint32), ("k_whir", c_uint32)]


class MatrixStruct(Structure):
    """``ovm_matrix_t``: one column-major device matrix."""

    _fields_ = [("d_ptr", c_void_p), ("width", c_uint64), ("height", c_uint64)]


class ProverConfigStruct(Structure):
    """``ovm_prover_config_t`` (reference: GpuProverConfig, crates/cuda-backend/src/device.rs:25-30)."""

    _fields_ = [("cache_rs_code_matrix", c_uint32), ("zerocheck_save_memory", c_uint32), ("round0_slab_log_rows", c_uint32),
                ("gkr_drop_leaves", c_uint32), ("cache_limit_bytes", c_uint64)]


class IpcStatusStruct(Structure):
    """ovm_ipc_status_t"""
    _fields_ = [("degraded", c_uint32), ("mappings", c_uint32), ("opens", c_uint64), ("exchanges", c_uint64),
                ("staged_exchanges", c_uint64), ("bounced_exchanges", c_uint64), ("reason", ctypes.c_char * 192)]


class SpongeState(Structure):
    """``ovm_sponge_t`` (reference: crates/cuda-backend/src/sponge.rs:39-48)."""

    _fields_ = [("state", c_uint32 * 16), ("absorb_idx", c_uint32), ("sample_idx", c_uint32)]


_LIB = None

# name -> (restype, argtypes); every symbol include/openvm_hip.h declares.
_u32p = POINTER(c_uint32)
_u64p = POINTER(c_uint64)
SIGNATURES = {
    "ovm_hip_abi_version": (c_uint32, []),
    "ovm_hip_thread_shutdown": (c_int, []),
    "ovm_hip_init": (c_int, [c_int, _u32p]),
    "ovm_hip_error_name": (c_char_p, [c_int]),
    "ovm_hip_profile_enable": (c_int, [c_int]),
    "ovm_hip_profile_report": (c_uint64, [c_char_p, c_uint64]),
    "ovm_hip_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "ovm_hip_free": (c_int, [c_void_p]),
    "ovm_hip_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "ovm_hip_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "ovm_hip_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "ovm_hip_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "ovm_hip_stream_create": (c_int, [POINTER(c_void_p)]),
    "ovm_hip_stream_destroy": (c_int, [c_void_p]),
    "ovm_hip_stream_synchronize": (c_int, [c_void_p]),
    "ovm_hip_eval_to_coeff": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint32, c_void_p]),
    "ovm_hip_ntt_dif": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint32, c_void_p]),
    "ovm_hip_ntt_dif_ext": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint32, c_void_p]),
    "ovm_hip_rs_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_uint32, c_uint32,
                                  c_uint32, c_void_p]),
    "ovm_hip_merkle_scratch_words": (c_uint64, [c_uint32, c_uint32]),
    "ovm_hip_merkle_layers_words": (c_uint64, [c_uint32, c_uint32]),
    "ovm_hip_merkle_tree": (c_int, [c_void_p, c_uint64, c_uint32, c_uint32, c_void_p, c_void_p,
                                    c_void_p]),
    "ovm_hip_merkle_tree_ext": (c_int, [c_void_p, c_uint64, c_uint32, c_uint32, c_void_p, c_void_p,
                                        c_void_p]),
    "ovm_hip_poseidon2_permute": (c_int, [c_void_p, c_uint64, c_void_p]),
    "ovm_hip_poseidon2_permute_rc": (c_int, [c_void_p, c_uint64, c_void_p, c_void_p]),
    "ovm_hip_sponge_grind": (c_int, [POINTER(SpongeState), c_uint32, _u32p, c_void_p]),
    "ovm_hip_gather_rows": (c_int, [c_void_p, c_uint64, c_uint64, c_uint32, c_void_p, c_uint64,
                                    c_void_p, c_void_p]),
    "ovm_hip_gather_digests": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p]),
    "ovm_hip_pk_load": (c_int, [c_void_p, c_uint64, POINTER(c_void_p), c_void_p]),
    "ovm_hip_pk_free": (None, [c_void_p]),
    "ovm_hip_pk_preprocessed_commitment": (c_int, [c_void_p, c_uint32, c_void_p]),
    "ovm_hip_staged_selftest": (c_int, [c_void_p, c_uint64, c_uint32, c_uint32, c_uint64, _u64p]),
    "ovm_hip_pk_preprocessed_data": (c_int, [c_void_p, c_uint32, c_void_p, _u64p, _u64p, c_void_p, c_void_p]),
    "ovm_hip_program_stats": (c_int, [c_void_p, c_uint64, c_uint32, c_uint32, c_void_p, c_void_p, c_void_p]),
    "ovm_hip_host_gkr_batch_selftest": (c_int, [c_uint32, c_uint64]),
    "ovm_hip_host_permute_selftest": (c_int, [c_uint32, c_uint64]),
    "ovm_hip_last_error": (c_char_p, []),
    "ovm_hip_trim": (c_int, []),
    "ovm_hip_set_prover_config": (c_int, [POINTER(ProverConfigStruct)]),
    "ovm_hip_memory_stats": (c_int, [_u64p, _u64p, _u64p, c_int]),
    "ovm_hip_ipc_comm_create": (c_int, [c_void_p, c_uint32, c_uint32, POINTER(c_void_p)]),
    "ovm_hip_ipc_comm_destroy": (None, [c_void_p]),
    "ovm_hip_ipc_allgather_host": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    "ovm_hip_ipc_allgather_device": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    "ovm_hip_ipc_alltoall_device": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    "ovm_hip_ipc_abort": (c_int, [c_void_p, c_int]),
    "ovm_hip_ipc_comm_status": (c_int, [c_void_p, c_void_p]),
    "ovm_hip_pool_stats": (c_int, [_u64p]),
    "ovm_hip_pool_selftest": (c_int, [c_uint64, c_uint32]),
    "ovm_hip_constraint_eval_buffer_size": (c_uint64, [c_void_p, c_uint32]),
    "ovm_hip_commit": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, POINTER(c_void_p), c_void_p]),
    "ovm_hip_commit_matrix": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64, c_void_p, POINTER(c_void_p), c_void_p]),
    "ovm_hip_commit_matrices": (c_int, [POINTER(PcsParams), c_uint32, c_void_p, c_void_p, POINTER(c_void_p), c_void_p]),
    "ovm_hip_proof_size": (c_int, [c_void_p, c_uint32, c_void_p, _u64p]),
    "ovm_hip_pcs_free": (None, [c_void_p]),
    "ovm_hip_prove_rap_constraints": (c_int, [c_void_p, c_uint32, c_void_p, POINTER(SpongeState), c_void_p, c_uint64,
                                              _u64p, _u64p, c_void_p, c_uint32, _u32p, c_void_p]),
    "ovm_hip_prove_openings": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_uint32,
                                       POINTER(SpongeState), c_void_p, c_uint64, _u64p, _u64p, c_void_p]),
    "ovm_hip_prove": (c_int, [c_void_p, c_uint32, c_void_p, POINTER(SpongeState), c_void_p, c_uint64, _u64p, _u64p,
                              c_void_p]),
    "ovm_hip_shm_comm_create": (c_int, [c_char_p, c_uint32, c_uint32, c_uint64, POINTER(c_void_p)]),
    "ovm_hip_shm_segment_bytes": (c_uint64, [c_uint32, c_uint64]),
    "ovm_hip_shm_comm_attach": (c_int, [c_void_p, c_uint32, c_uint32, c_uint64, POINTER(c_void_p)]),
    "ovm_hip_shm_comm_destroy": (None, [c_void_p]),
    "ovm_hip_shm_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64]),
    "ovm_hip_shm_comm_abort": (c_int, [c_void_p, c_int]),
    "ovm_hip_shm_comm_poisoned": (c_int, [c_void_p]),
    "ovm_hip_shard_last_error": (ctypes.c_char_p, []),
    "ovm_hip_gkr_shard_selftest": (c_int, [c_uint32, c_uint32, _u64p, c_uint64, c_int]),
    "ovm_hip_selftest_error": (c_char_p, []),
    "ovm_hip_shard_exchange_selftest": (c_int, [c_void_p, c_uint64, c_uint32, c_uint64]),
    "ovm_hip_prove_sharded": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, _u32p, POINTER(SpongeState), c_void_p, c_uint64,
                                      _u64p, _u64p, c_void_p]),
}


def _lib():
    """Loads libopenvm_hip.so. Raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with stark-backend_amd/build.sh "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        # PyTorch ships its own HIP runtime; load it first so that this library binds to the same
        # libamdhip64 (two runtimes in one process do not see each other's devices).
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI drifted from the header
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def lib():
    return _lib()


def check(code: int, what: str) -> None:
    if code != 0:
        raise OvmError(code, what)


# ---------------------------------------------------------------------------------------------
# Field helpers (host, numpy): the ABI carries Montgomery words.
# ---------------------------------------------------------------------------------------------
_RINV = pow(1 << MONTY_BITS, P - 2, P)


def to_mont(x) -> np.ndarray:
    """canonical u32 -> Montgomery u32 (x * 2^32 mod P)."""
    a = np.asarray(x, dtype=np.uint64) % np.uint64(P)
    return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


def from_mont(x) -> np.ndarray:
    """Montgomery u32 -> canonical u32."""
    a = np.asarray(x, dtype=np.uint64)
    # a * RINV can exceed 64 bits: split RINV into 16-bit halves
    lo = np.uint64(_RINV & 0xFFFF)
    hi = np.uint64(_RINV >> 16)
    r = (a * lo) % np.uint64(P)
    r = (r + (((a * hi) % np.uint64(P)) << np.uint64(16)) % np.uint64(P)) % np.uint64(P)
    return r.astype(np.uint32)


def bitrev(x, bits: int):
    """Bit-reverse the low `bits` bits of an int or integer array."""
    x = np.asarray(x, dtype=np.uint64)
    r = np.zeros_like(x)
    for i in range(bits):
        r |= ((x >> np.uint64(i)) & np.uint64(1)) << np.uint64(bits - 1 - i)
    return r


# ---------------------------------------------------------------------------------------------
# Device handle
# ---------------------------------------------------------------------------------------------
class Device:
    """One HIP device + stream; owner of the twiddle tables (reference: GpuDevice,
    crates/cuda-backend/src/device.rs:58-79)."""

    def __init__(self, index: int = 0):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible")
        self.index = index
        self.torch = torch
        torch.cuda.set_device(index)
        cu = c_uint32(0)
        check(_lib().ovm_hip_init(index, ctypes.byref(cu)), "ovm_hip_init")
        self.cu_count = cu.value

    # -- memory -------------------------------------------------------------------------
    @property
    def stream(self):
        return c_void_p(self.torch.cuda.current_stream(self.index).cuda_stream)

    def empty(self, n_words: int):
        return self.torch.empty(int(n_words), dtype=self.torch.int32, device=f"cuda:{self.index}")

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        if a.dtype == np.uint32:
            return self.torch.from_numpy(a.view(np.int32).reshape(-1)).to(f"cuda:{self.index}")
        if a.dtype == np.uint64:
            return self.torch.from_numpy(a.view(np.int64).reshape(-1)).to(f"cuda:{self.index}")
        raise TypeError(a.dtype)

    @staticmethod
    def download(t) -> np.ndarray:
        a = t.cpu().numpy()
        return a.view(np.uint32 if a.dtype == np.int32 else np.uint64)

    def sync(self):
        self.torch.cuda.synchronize(self.index)

    # -- kernels ------------------------------------------------------------------------
    def eval_to_coeff(self, d_in, d_out, width, log_height, l_skip):
        check(_lib().ovm_hip_eval_to_coeff(d_in.data_ptr(), d_out.data_ptr(), width, log_height,
                                           l_skip, self.stream), "ovm_hip_eval_to_coeff")

    def ntt_dif(self, d_in, d_out, width, log_n, log_n_in=None, ext=False):
        fn = _lib().ovm_hip_ntt_dif_ext if ext else _lib().ovm_hip_ntt_dif
        check(fn(d_in.data_ptr(), d_out.data_ptr(), width, log_n,
                 log_n if log_n_in is None else log_n_in, self.stream), "ovm_hip_ntt_dif")

    def rs_encode(self, d_stacked, width, log_height, l_skip, log_blowup, d_scratch=None):
        if d_scratch is None:
            d_scratch = self.empty(width << log_height)
        d_code = self.empty(width << (log_height + log_blowup))
        check(_lib().ovm_hip_rs_encode(d_stacked.data_ptr(), d_scratch.data_ptr(), d_code.data_ptr(),
                                       width, log_height, l_skip, log_blowup, self.stream),
              "ovm_hip_rs_encode")
        return d_code

    def merkle_tree(self, d_matrix, width, log_rows, log_rows_per_query, ext=False):
        scratch = self.empty(max(1, _lib().ovm_hip_merkle_scratch_words(log_rows, log_rows_per_query)))
        layers = self.empty(_lib().ovm_hip_merkle_layers_words(log_rows, log_rows_per_query))
        fn = _lib().ovm_hip_merkle_tree_ext if ext else _lib().ovm_hip_merkle_tree
        check(fn(d_matrix.data_ptr(), width, log_rows, log_rows_per_query, scratch.data_ptr(),
                 layers.data_ptr(), self.stream), "ovm_hip_merkle_tree")
        return layers

    def poseidon2_permute(self, d_states, n):
        check(_lib().ovm_hip_poseidon2_permute(d_states.data_ptr(), n, self.stream),
              "ovm_hip_poseidon2_permute")

    def poseidon2_permute_rc(self, d_states, n, d_rc141):
        check(_lib().ovm_hip_poseidon2_permute_rc(d_states.data_ptr(), n, d_rc141.data_ptr(), self.stream),
              "ovm_hip_poseidon2_permute_rc")

    def grind(self, sponge: SpongeState, bits: int) -> int:
        w = c_uint32(0)
        check(_lib().ovm_hip_sponge_grind(ctypes.byref(sponge), bits, ctypes.byref(w), self.stream),
              "ovm_hip_sponge_grind")
        return w.value


# ---------------------------------------------------------------------------------------------
# Proving key + prove (mirrors StarkEngine::prove / Coordinator::prove on resident traces)
# ---------------------------------------------------------------------------------------------
class ProvingKey:
    """Device proving key built from the flat u32 blob (see include/openvm_hip.h)."""

    def __init__(self, blob: np.ndarray, stream=None):
        b = np.ascontiguousarray(blob, dtype=np.uint32)
        h = c_void_p(0)
        rc = _lib().ovm_hip_pk_load(b.ctypes.data, b.size, ctypes.byref(h), stream)
        if rc != 0:
            msg = _lib().ovm_hip_last_error()
            raise OvmError(rc, f"ovm_hip_pk_load [{msg.decode() if msg else ''}]")
        self.handle = h

    def preprocessed_commitment(self, air_idx: int) -> np.ndarray:
        out = np.zeros(8, dtype=np.uint32)
        check(_lib().ovm_hip_pk_preprocessed_commitment(self.handle, air_idx, out.ctypes.data),
              "ovm_hip_pk_preprocessed_commitment")
        return out

    def __del__(self):
        try:
            if self.handle:
                _lib().ovm_hip_pk_free(self.handle)
                self.handle = None
        except Exception:
            pass


class PcsData:
    """Opaque ``ovm_pcs_t`` handle (``Device.commit_traces`` / ``Device.commit_matrix``)."""

    def __init__(self, handle, root, keep=None):
        self.handle = handle
        self.root = root  # 8 Montgomery words
        self._keep = keep  # the committed tensor: the handle points into it

    def free(self):
        if self.handle:
            _lib().ovm_hip_pcs_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _trace_structs(traces):
    """traces: list of (air_idx, device int32 tensor [column-major Montgomery], height, public values
    as canonical ints [, list of PcsData for the cached main partitions]). Returns (array, keepalive)."""
    n = len(traces)
    arr = (TraceStruct * n)()
    keep = []
    for i, t in enumerate(traces):
        pv = to_mont(np.array(t[3], dtype=np.uint32)) if len(t[3]) else np.zeros(1, np.uint32)
        cached = list(t[4]) if len(t) > 4 and t[4] is not None else []
        handles = (c_void_p * max(len(cached), 1))(*[c.handle for c in cached])
        keep += [pv, handles, cached]
        arr[i].air_idx = t[0]
        arr[i].num_public_values = len(t[3])
        arr[i].d_common_main = t[1].data_ptr() if t[1] is not None else None
        arr[i].height = t[2]
        arr[i].public_values = pv.ctypes.data
        arr[i].num_cached = len(cached)
        arr[i].cached = ctypes.cast(handles, c_void_p)
    return arr, keep


def _fail(rc, what):
    msg = _lib().ovm_hip_last_error()
    raise OvmError(rc, f"{what} [{msg.decode() if msg else ''}]")


def proof_size(pk: ProvingKey, shapes) -> list:
    """Section end offsets (header, GKR, batch constraints, stacking, WHIR) of the proof for traces of the
    given (air_idx, height) shapes; the last one is the proof length. Host only."""
    arr = (TraceStruct * len(shapes))()
    for i, (air_idx, height) in enumerate(shapes):
        arr[i].air_idx, arr[i].height = air_idx, height
    ends = (c_uint64 * 5)()
    rc = _lib().ovm_hip_proof_size(pk.handle, len(shapes), arr, ends)
    if rc != 0:
        _fail(rc, "ovm_hip_proof_size")
    return [int(x) for x in ends]


def _prove(self, pk: ProvingKey, traces, sponge: SpongeState | None = None, proof_cap: int | None = None):
    """Coordinator::prove on resident traces. Returns (proof bytes, section end offsets). The buffer is
    sized by ovm_hip_proof_size unless `proof_cap` is given."""
    arr, keep = _trace_structs(traces)
    if proof_cap is None:
        proof_cap = proof_size(pk, [(t[0], t[2]) for t in traces])[4]
    out = np.zeros(proof_cap, dtype=np.uint8)
    plen = c_uint64(0)
    secs = (c_uint64 * 5)()
    rc = _lib().ovm_hip_prove(pk.handle, len(traces), arr, ctypes.byref(sponge) if sponge is not None else None,
                              out.ctypes.data, proof_cap, ctypes.byref(plen), secs, self.stream)
    if rc != 0:
        _fail(rc, "ovm_hip_prove")
    return out[:plen.value].tobytes(), [int(x) for x in secs]


Device.prove = _prove


# ---- stage-level entries (TraceCommitter / MultiRapProver / OpeningProver, hal.rs:84-138) ------
def _commit_traces(self, pk: ProvingKey, traces) -> PcsData:
    """TraceCommitter::commit of the common mains."""
    arr, keep = _trace_structs(traces)
    root = np.zeros(8, dtype=np.uint32)
    h = c_void_p(0)
    rc = _lib().ovm_hip_commit(pk.handle, len(traces), arr, root.ctypes.data, ctypes.byref(h), self.stream)
    if rc != 0:
        _fail(rc, "ovm_hip_commit")
    return PcsData(h, root, keep=[t[1] for t in traces])


def _commit_matrix(self, pk: ProvingKey, tensor, width: int, height: int) -> PcsData:
    """Single-matrix commitment (a cached main partition)."""
    root = np.zeros(8, dtype=np.uint32)
    h = c_void_p(0)
    rc = _lib().ovm_hip_commit_matrix(pk.handle, tensor.data_ptr(), width, height, root.ctypes.data, ctypes.byref(h),
                                      self.stream)
    if rc != 0:
        _fail(rc, "ovm_hip_commit_matrix")
    return PcsData(h, root, keep=tensor)


def _commit_matrices(self, params: dict, mats) -> PcsData:
    """TraceCommitter::commit(&[&Matrix]) (hal.rs:84-87): `mats` = [(tensor, width, height)], sorted by
    height descending; `params` needs l_skip, n_stack, log_blowup, k_whir."""
    pp = PcsParams(params["l_skip"], params["n_stack"], params["log_blowup"], params["k_whir"])
    arr = (MatrixStruct * len(mats))()
    for i, (t, w, h) in enumerate(mats):
        arr[i].d_ptr, arr[i].width, arr[i].height = t.data_ptr(), w, h
    root = np.zeros(8, dtype=np.uint32)
    h = c_void_p(0)
    rc = _lib().ovm_hip_commit_matrices(ctypes.byref(pp), len(mats), arr, root.ctypes.data, ctypes.byref(h), self.stream)
    if rc != 0:
        _fail(rc, "ovm_hip_commit_matrices")
    return PcsData(h, root, keep=[m[0] for m in mats])


# ---- sharded proving: one joint proof over the ranks of one node (include/openvm_hip.h) ----
class ShardStruct(Structure):
    """``ovm_shard_t``."""

    _fields_ = [("rank", c_uint32), ("world", c_uint32), ("ctx", c_void_p), ("allgather_host", c_void_p),
                ("allgather_device", c_void_p), ("alltoall_device", c_void_p), ("abort", c_void_p)]


_AG_DEVICE = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p)


class _DevPtr:
    """A raw device range as a ``__cuda_array_interface__`` object (so torch can view it without a copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class Shard:
    """This rank's communicator for ``Device.prove_sharded``: the library's shared-memory exchange for the
    small per-round values, and ``torch.distributed`` (RCCL over xGMI with the nccl backend; host-staged with
    gloo, which is what the CPU tests and several ranks sharing ONE GPU use) for the one bulk all-gather.
    ``name``: the same unique shared-memory name on every rank (e.g. from rank 0 via broadcast_object_list)."""

    def __init__(self, rank: int, world: int, name: str, slot_bytes: int = 1 << 20, group=None, host_buffers: bool = False,
                 bulk: str = "torch", alltoall: bool = True):
        import torch
        import torch.distributed as dist

        self.rank, self.world, self.group = rank, world, group
        self.host_buffers = host_buffers  # the bulk exchange is handed HOST pointers (CPU tests of the callback)
        # bulk = "torch": torch.distributed (nccl = RCCL; gloo = host-staged). bulk = "ipc": the library's own exchange (peer
        # buffers mapped through HIP IPC, device-to-device pulls; no process group needed). bulk = "shm": staged through the library's own
        # shared-memory exchange, no process group at all -- ranks that share one GPU in the tests.
        self.bulk = bulk
        h = c_void_p(0)
        check(_lib().ovm_hip_shm_comm_create(name.encode(), rank, world, slot_bytes, ctypes.byref(h)), "ovm_hip_shm_comm_create")
        self.comm = h
        self._torch, self._dist = torch, dist
        self._cb = _AG_DEVICE(self._allgather_device)
        self._cb_a2a = _AG_DEVICE(self._alltoall_device)
        # bulk = "ipc": the library's own exchange (IPC-mapped peer buffers + device-to-device pulls, csrc/ipc_comm.cpp)
        self.ipc = c_void_p(0)
        check(_lib().ovm_hip_ipc_comm_create(h, rank, world, ctypes.byref(self.ipc)), "ovm_hip_ipc_comm_create")
        self._alltoall = alltoall
        self._set_abi()
        self.bulk_bytes = 0
        self.alltoall_calls = 0

    def ipc_status(self) -> dict:
        """ovm_hip_ipc_comm_status: is the library's IPC exchange on mapped peer buffers or degraded to host staging, and why."""
        st = IpcStatusStruct()
        check(_lib().ovm_hip_ipc_comm_status(self.ipc, ctypes.byref(st)), "ovm_hip_ipc_comm_status")
        return {"degraded": bool(st.degraded), "mappings": int(st.mappings), "opens": int(st.opens), "exchanges": int(st.exchanges),
                "staged_exchanges": int(st.staged_exchanges), "bounced_exchanges": int(st.bounced_exchanges), "reason": st.reason.decode(errors="replace")}

    def _set_abi(self):
        """(Re)build the ovm_shard_t for the current ``bulk`` mode. alltoall=False hands the library a null all-to-all:
        the mu-batch vectors are then all-gathered whole."""
        lib = _lib()
        if self.bulk == "ipc" and not self.host_buffers:  # no Python in the data path at all
            self.abi = ShardStruct(self.rank, self.world, self.ipc, ctypes.cast(lib.ovm_hip_ipc_allgather_host, c_void_p),
                                   ctypes.cast(lib.ovm_hip_ipc_allgather_device, c_void_p),
                                   ctypes.cast(lib.ovm_hip_ipc_alltoall_device, c_void_p) if self._alltoall else None,
                                   ctypes.cast(lib.ovm_hip_ipc_abort, c_void_p))
        else:
            self.abi = ShardStruct(self.rank, self.world, self.comm, ctypes.cast(lib.ovm_hip_shm_allgather, c_void_p),
                                   ctypes.cast(self._cb, c_void_p), ctypes.cast(self._cb_a2a, c_void_p) if self._alltoall else None,
                                   ctypes.cast(lib.ovm_hip_shm_comm_abort, c_void_p))

    def _alltoall_device(self, ctx, d_send, d_recv, nbytes, stream):
        """d_recv[s * nbytes ..] = rank s's d_send[rank * nbytes ..] (``nbytes`` per PAIR of ranks)."""
        try:
            torch, dist = self._torch, self._dist
            total = nbytes * self.world
            self.alltoall_calls += 1
            self.bulk_bytes += int(total)
            if self.host_buffers:
                send = torch.frombuffer((ctypes.c_uint8 * total).from_address(d_send), dtype=torch.uint8)
                recv = torch.empty(total, dtype=torch.uint8)
                self._host_alltoall(recv, send, nbytes)
                ctypes.memmove(d_recv, recv.data_ptr(), total)
                return 0
            torch.cuda.current_stream().synchronize()
            check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
            if self.bulk == "ipc":
                return int(_lib().ovm_hip_ipc_alltoall_device(self.ipc, d_send, d_recv, nbytes, stream))
            if self.bulk == "shm":  # (tests: ranks sharing one GPU, no process group) every rank's whole vector, then pick
                send = np.empty(total, dtype=np.uint8)
                everything = np.empty(total * self.world, dtype=np.uint8)
                check(_lib().ovm_hip_memcpy_d2h(send.ctypes.data, d_send, total, stream), "ovm_hip_memcpy_d2h")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                check(_lib().ovm_hip_shm_allgather(self.comm, send.ctypes.data, everything.ctypes.data, total), "ovm_hip_shm_allgather")
                recv = np.ascontiguousarray(everything.reshape(self.world, self.world, nbytes)[:, self.rank, :])
                check(_lib().ovm_hip_memcpy_h2d(d_recv, recv.ctypes.data, total, stream), "ovm_hip_memcpy_h2d")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                return 0
            if dist.get_backend(self.group) == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
                try:
                    send = torch.as_tensor(_DevPtr(d_send, total), device=dev)
                    recv = torch.as_tensor(_DevPtr(d_recv, total), device=dev)
                    staged = False
                except Exception:
                    send = torch.empty(total, dtype=torch.uint8, device=dev)
                    recv = torch.empty(total, dtype=torch.uint8, device=dev)
                    check(_lib().ovm_hip_memcpy_d2d(send.data_ptr(), d_send, total, stream), "ovm_hip_memcpy_d2d")
                    check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                    staged = True
                dist.all_to_all_single(recv, send, group=self.group)  # equal splits of nbytes
                if staged:
                    torch.cuda.current_stream().synchronize()
                    check(_lib().ovm_hip_memcpy_d2d(d_recv, recv.data_ptr(), total, stream), "ovm_hip_memcpy_d2d")
                    check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
            else:  # host-staged (gloo)
                send = torch.empty(total, dtype=torch.uint8)
                recv = torch.empty(total, dtype=torch.uint8)
                check(_lib().ovm_hip_memcpy_d2h(send.data_ptr(), d_send, total, stream), "ovm_hip_memcpy_d2h")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                self._host_alltoall(recv, send, nbytes)
                check(_lib().ovm_hip_memcpy_h2d(d_recv, recv.data_ptr(), total, stream), "ovm_hip_memcpy_h2d")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
            torch.cuda.current_stream().synchronize()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import sys
            print(f"[ovm] alltoall_device failed on rank {self.rank}: {e!r}", file=sys.stderr, flush=True)
            return 999

    def _host_alltoall(self, recv, send, nbytes):
        """CPU tensors: gloo has no all_to_all on every build, so gather everything and pick this rank's slices."""
        dist = self._dist
        parts = [self._torch.empty(send.numel(), dtype=self._torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, send, group=self.group)
        for s, part in enumerate(parts):
            recv[s * nbytes:(s + 1) * nbytes] = part[self.rank * nbytes:(self.rank + 1) * nbytes]

    def _allgather_device(self, ctx, d_mine, d_all, nbytes, stream):
        try:
            torch, dist = self._torch, self._dist
            if self.host_buffers:
                send = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(d_mine), dtype=torch.uint8)
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, send, group=self.group)
                for r, part in enumerate(parts):
                    ctypes.memmove(d_all + r * nbytes, part.data_ptr(), nbytes)
                self.bulk_bytes += int(nbytes) * self.world
                return 0
            # what the prover enqueued before the exchange: its stream is torch's current stream in this package, but the
            # collective must not start early if a caller hands the prover another one
            torch.cuda.current_stream().synchronize()
            check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
            self.bulk_bytes += int(nbytes) * self.world
            if self.bulk == "ipc":
                return int(_lib().ovm_hip_ipc_allgather_device(self.ipc, d_mine, d_all, nbytes, stream))
            if self.bulk == "shm":
                send = np.empty(nbytes, dtype=np.uint8)
                recv = np.empty(nbytes * self.world, dtype=np.uint8)
                check(_lib().ovm_hip_memcpy_d2h(send.ctypes.data, d_mine, nbytes, stream), "ovm_hip_memcpy_d2h")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                check(_lib().ovm_hip_shm_allgather(self.comm, send.ctypes.data, recv.ctypes.data, nbytes), "ovm_hip_shm_allgather")
                check(_lib().ovm_hip_memcpy_h2d(d_all, recv.ctypes.data, nbytes * self.world, stream), "ovm_hip_memcpy_h2d")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                return 0
            if dist.get_backend(self.group) == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
                try:  # views of the prover's own buffers: no copy
                    send = torch.as_tensor(_DevPtr(d_mine, nbytes), device=dev)
                    recv = torch.as_tensor(_DevPtr(d_all, nbytes * self.world), device=dev)
                    staged = False
                except Exception:  # a torch build without __cuda_array_interface__ views: stage through torch tensors
                    send = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    recv = torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev)
                    check(_lib().ovm_hip_memcpy_d2d(send.data_ptr(), d_mine, nbytes, stream), "ovm_hip_memcpy_d2d")
                    check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                    staged = True
                dist.all_gather_into_tensor(recv, send, group=self.group)
                if staged:
                    torch.cuda.current_stream().synchronize()
                    check(_lib().ovm_hip_memcpy_d2d(d_all, recv.data_ptr(), nbytes * self.world, stream), "ovm_hip_memcpy_d2d")
                    check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
            else:  # host-staged (gloo): the same call sequence without a device collective
                send = torch.empty(nbytes, dtype=torch.uint8)
                check(_lib().ovm_hip_memcpy_d2h(send.data_ptr(), d_mine, nbytes, stream), "ovm_hip_memcpy_d2h")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, send, group=self.group)
                for r, part in enumerate(parts):
                    check(_lib().ovm_hip_memcpy_h2d(d_all + r * nbytes, part.data_ptr(), nbytes, stream), "ovm_hip_memcpy_h2d")
                check(_lib().ovm_hip_stream_synchronize(stream), "ovm_hip_stream_synchronize")
            torch.cuda.current_stream().synchronize()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import sys
            print(f"[ovm] allgather_device failed on rank {self.rank}: {e!r}", file=sys.stderr, flush=True)
            return 999

    def check_bulk(self, stream=None) -> bool:
        """Pre-flight of the device exchanges: a small all-gather and all-to-all of rank-dependent bytes through the very
        callbacks the prover will call, verified on every rank; the ranks then agree (one byte each through the
        shared-memory exchange) on whether ALL of them saw the right data. On False the communicator has switched itself
        to the next exchange in line -- bulk="ipc" (the library's own: IPC-mapped peer buffers + device-to-device pulls),
        checked the same way, then bulk="shm" (host-staged through the shared-memory segment: slow, needs nothing but
        this library) -- so a deployment whose collective library misbehaves still proves."""
        torch = self._torch
        if self.host_buffers or self.world == 1:
            return True
        dev = torch.device("cuda", torch.cuda.current_device())
        sp = stream if stream is not None else c_void_p(torch.cuda.current_stream().cuda_stream)
        n, W, ok = 1 << 16, self.world, True

        def pattern(a, b):
            return ((torch.arange(n, dtype=torch.int64) * (2 * a + 3) + 7 * b + 1) % 251).to(torch.uint8)
        try:
            send = pattern(self.rank, 0).to(dev)
            recv = torch.zeros(n * W, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ok &= self._allgather_device(None, send.data_ptr(), recv.data_ptr(), n, sp) == 0
            torch.cuda.synchronize()
            ok &= bool(torch.equal(recv.cpu(), torch.cat([pattern(r, 0) for r in range(W)])))
            send2 = torch.cat([pattern(self.rank, s + 1) for s in range(W)]).to(dev)
            recv2 = torch.zeros(n * W, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ok &= self._alltoall_device(None, send2.data_ptr(), recv2.data_ptr(), n, sp) == 0
            torch.cuda.synchronize()
            ok &= bool(torch.equal(recv2.cpu(), torch.cat([pattern(s, self.rank + 1) for s in range(W)])))
        except Exception:  # noqa: BLE001
            ok = False
        mine = np.array([1 if ok else 0], dtype=np.uint8)
        every = np.zeros(W, dtype=np.uint8)
        check(_lib().ovm_hip_shm_allgather(self.comm, mine.ctypes.data, every.ctypes.data, 1), "ovm_hip_shm_allgather")
        self.bulk_bytes, self.alltoall_calls = 0, 0
        if not every.all():
            # next in line: the library's own IPC exchange, then host staging through the shared-memory segment
            self.bulk = "ipc" if self.bulk == "torch" else "shm"
            self._set_abi()
            if self.bulk == "ipc":
                self.check_bulk(stream)
            return False
        return True

    @property
    def poisoned(self) -> bool:
        """A joint call over this communicator timed out or was aborted by a peer: it must be closed and recreated."""
        return bool(self.comm) and bool(_lib().ovm_hip_shm_comm_poisoned(self.comm))

    def close(self):
        if self.ipc:
            _lib().ovm_hip_ipc_comm_destroy(self.ipc)
            self.ipc = None
        if self.comm:
            _lib().ovm_hip_shm_comm_destroy(self.comm)
            self.comm = None


def shard_exchange_selftest(shard: Shard, n: int, rounds: int, seed: int = 1) -> None:
    rc = _lib().ovm_hip_shard_exchange_selftest(ctypes.byref(shard.abi), n, rounds, seed)
    if rc != 0:
        raise OvmError(rc, f"ovm_hip_shard_exchange_selftest [{_lib().ovm_hip_selftest_error().decode()}]")


def _prove_sharded(self, pk: ProvingKey, shard: Shard, traces, owners, sponge: SpongeState | None = None):
    """ONE joint proof over ``shard.world`` ranks. ``traces`` lists every trace of the proof as in ``prove``;
    entries owned by another rank may carry ``None`` for the tensor and the cached list. Returns
    (proof bytes, section ends) on rank 0 and (b"", []) elsewhere."""
    arr, keep = _trace_structs(traces)
    own = (c_uint32 * len(traces))(*[int(o) for o in owners])
    cap = proof_size(pk, [(t[0], t[2]) for t in traces])[4] if shard.rank == 0 else 0
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    plen = c_uint64(0)
    secs = (c_uint64 * 5)()
    rc = _lib().ovm_hip_prove_sharded(pk.handle, ctypes.byref(shard.abi), len(traces), arr, own,
                                      ctypes.byref(sponge) if sponge is not None else None, out.ctypes.data, cap,
                                      ctypes.byref(plen), secs, self.stream)
    if rc != 0:
        _fail(rc, "ovm_hip_prove_sharded")
    if shard.rank != 0:
        return b"", []
    return out[:plen.value].tobytes(), [int(x) for x in secs]


def _prove_rap_constraints(self, pk: ProvingKey, traces, sponge: SpongeState, cap: int = 1 << 24):
    """MultiRapProver::prove_rap_constraints. Returns (GkrProof bytes, BatchConstraintProof bytes,
    r as an (n, 4) array of Montgomery words); ``sponge`` is updated in place."""
    arr, keep = _trace_structs(traces)
    out = np.zeros(cap, dtype=np.uint8)
    gkr_len, total = c_uint64(0), c_uint64(0)
    r = np.zeros(4 * 256, dtype=np.uint32)
    r_len = c_uint32(0)
    rc = _lib().ovm_hip_prove_rap_constraints(pk.handle, len(traces), arr, ctypes.byref(sponge), out.ctypes.data, cap,
                                              ctypes.byref(gkr_len), ctypes.byref(total), r.ctypes.data, 256,
                                              ctypes.byref(r_len), self.stream)
    if rc != 0:
        _fail(rc, "ovm_hip_prove_rap_constraints")
    b = out[:total.value].tobytes()
    return b[:gkr_len.value], b[gkr_len.value:], r[:4 * r_len.value].reshape(-1, 4).copy()


def _prove_openings(self, pk: ProvingKey, traces, pcs: PcsData, r: np.ndarray, sponge: SpongeState,
                    cap: int = 1 << 26):
    """OpeningProver::prove_openings. Returns (StackingProof bytes, WhirProof bytes)."""
    arr, keep = _trace_structs(traces)
    rr = np.ascontiguousarray(r, dtype=np.uint32).reshape(-1)
    out = np.zeros(cap, dtype=np.uint8)
    st_len, total = c_uint64(0), c_uint64(0)
    rc = _lib().ovm_hip_prove_openings(pk.handle, len(traces), arr, pcs.handle, rr.ctypes.data, rr.size // 4,
                                       ctypes.byref(sponge), out.ctypes.data, cap, ctypes.byref(st_len),
                                       ctypes.byref(total), self.stream)
    if rc != 0:
        _fail(rc, "ovm_hip_prove_openings")
    b = out[:total.value].tobytes()
    return b[:st_len.value], b[st_len.value:]


Device.prove_sharded = _prove_sharded
Device.commit_matrix = _commit_matrix
Device.commit_matrices = _commit_matrices
Device.commit_traces = _commit_traces
Device.prove_rap_constraints = _prove_rap_constraints
Device.prove_openings = _prove_openings


def gkr_shard_selftest(world: int, TR: int, ends, seed: int = 1, three_points: bool = False) -> None:
    """Host-only: the sharded LogUp-GKR driver with `world` simulated ranks owning the leaf runs that end at
    `ends`, against the unsharded run (csrc/selftest.cpp). Raises on a mismatch."""
    e = (c_uint64 * world)(*[int(x) for x in ends])
    rc = _lib().ovm_hip_gkr_shard_selftest(world, TR, e, seed, 1 if three_points else 0)
    if rc != 0:
        raise OvmError(rc, f"ovm_hip_gkr_shard_selftest [{_lib().ovm_hip_selftest_error().decode()}]")


def program_stats(blob, air_idx: int = 0, kind: int = 0) -> dict:
    """{instrs, slots, remat_limit} of the register program AIR ``air_idx`` compiles to (kind 0 zerocheck, 1 LogUp, 2 leaves)."""
    b = np.ascontiguousarray(blob, dtype=np.uint32)
    out = (c_uint32 * 3)()
    rc = _lib().ovm_hip_program_stats(b.ctypes.data, b.size, air_idx, kind, ctypes.byref(out, 0), ctypes.byref(out, 4), ctypes.byref(out, 8))
    if rc != 0:
        _fail(rc, "ovm_hip_program_stats")
    return {"instrs": int(out[0]), "slots": int(out[1]), "remat_limit": int(out[2])}


def staged_selftest(blob, stage_rows: int, acc_rows: int, seed: int = 1) -> int:
    """Stages every constraint program of the key (csrc/program.h stage_program) with the given layout
    and runs it on a random row next to the plain interpreter, on the host; no GPU needed. Returns the
    number of blocks staged; raises OvmError on a mismatch."""
    b = np.ascontiguousarray(blob, dtype=np.uint32)
    blocks = c_uint64(0)
    rc = _lib().ovm_hip_staged_selftest(b.ctypes.data, b.size, stage_rows, acc_rows, seed, ctypes.byref(blocks))
    if rc != 0:
        _fail(rc, "ovm_hip_staged_selftest")
    return int(blocks.value)


def set_prover_config(cache_rs_code_matrix: bool = True, zerocheck_save_memory: bool = False,
                      round0_slab_log_rows: int = 0, cache_limit_bytes: int = 0, gkr_drop_leaves: bool = False) -> None:
    """Memory-bounded modes of the calling thread's provers (include/openvm_hip.h: ovm_prover_config_t)."""
    cfg = ProverConfigStruct(int(cache_rs_code_matrix), int(zerocheck_save_memory), round0_slab_log_rows, int(gkr_drop_leaves),
                             cache_limit_bytes)
    check(_lib().ovm_hip_set_prover_config(ctypes.byref(cfg)), "ovm_hip_set_prover_config")


def memory_stats(reset_peak: bool = False) -> dict:
    """Device bytes handed out by the library's pool (live / peak, process-wide), free in this thread's pool (cached),
    held from the driver in segments (now / peak), hipMalloc calls ever made, blocks handed out now."""
    live, peak, cached = c_uint64(0), c_uint64(0), c_uint64(0)
    _lib().ovm_hip_memory_stats(ctypes.byref(live), ctypes.byref(peak), ctypes.byref(cached), 1 if reset_peak else 0)
    pool = (c_uint64 * 4)()
    _lib().ovm_hip_pool_stats(pool)
    return {"live": live.value, "peak": peak.value, "cached": cached.value, "segments": int(pool[0]),
            "peak_segments": int(pool[1]), "driver_allocs": int(pool[2]), "live_blocks": int(pool[3])}


def pool_selftest(seed: int = 1, ops: int = 20000) -> None:
    """Device-free stress of the memory pool over host memory (csrc/block_pool.h)."""
    rc = _lib().ovm_hip_pool_selftest(seed, ops)
    if rc != 0:
        raise OvmError(rc, f"ovm_hip_pool_selftest [{_lib().ovm_hip_selftest_error().decode()}]")


def constraint_eval_buffer_size(pk: ProvingKey, air_idx: int) -> int:
    return int(_lib().ovm_hip_constraint_eval_buffer_size(pk.handle, air_idx))


def profile_enable(on) -> None:
    """False/0: off; True/1: every stage; 2: only the dominant kernel's label (``merkle.hash_rows``) -- one event pair
    per commitment, what bench.py keeps on inside its timed region."""
    _lib().ovm_hip_profile_enable(int(on))


def profile_report() -> dict:
    """{label: (calls, total_ms)} measured with HIP events on the launching stream."""
    buf = ctypes.create_string_buffer(1 << 16)
    _lib().ovm_hip_profile_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        label, calls, ms, _nested, _side = line.rsplit(" ", 4)
        out[label] = (int(calls), float(ms))
    return out


def profile_report_detail() -> dict:
    """{label: {calls, ms, nested_ms, side_ms}}: nested_ms was measured inside another open scope (whir.rs_commit contains
    Merkle and NTT labels), side_ms on the library's side stream; ms - nested_ms - side_ms is additive over the labels."""
    buf = ctypes.create_string_buffer(1 << 16)
    _lib().ovm_hip_profile_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        label, calls, ms, nested, side = line.rsplit(" ", 4)
        out[label] = {"calls": int(calls), "ms": float(ms), "nested_ms": float(nested), "side_ms": float(side)}
    return out


def max_over_ranks(value: float, device: str = "cpu") -> float:
    """Whole-job time of a multi-process run: MAX over ranks (all-reduce on the initialised
    torch.distributed group -- RCCL on GPUs, gloo on CPU); identity when not distributed."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
