"""longfellow-zk_amd -- MI355X-native prover hot path for longfellow-zk.

Host-side mirror (Python/ctypes) of the C ABI in include/lfgpu.h.  The compute
lives in hand-written HIP kernels (csrc/*.hip, built into liblfgpu.so by
__graft_entry__.build()).  There is NO CPU fallback: if the library is missing
or the GPU is absent every call raises.

The directory name contains a hyphen, so import it through
``__graft_entry__.load_package()`` (importlib) rather than ``import``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFGPU_LIB") or os.path.join(_HERE, "liblfgpu.so")  # LFGPU_LIB: an instrumented build of the same sources

FIELD_GF2_128 = 4  # FieldID, reference lib/proto/circuit_io.h:24-36
FIELD_FP128 = 6
FIELD_P256 = 1  # Fp256Base, 32-byte elements

# 2^32-order root of unity of Fp128 (reference lib/algebra/fp_p128.h:48-56), canonical value
FP128_OMEGA32 = 164956748514267535023998284330560247862
FP128_P = 2**128 - 2**108 + 1
# F64 = Fp<1>, p = 2^64 - 2^32 + 1, and its root of unity of order 2^32 (reference lib/algebra/fft_test.cc:208-213)
F64_P = 2**64 - 2**32 + 1
F64_OMEGA32 = 2752994695033296049

ABI_SYMBOLS = [
    "lfgpu_init", "lfgpu_shutdown", "lfgpu_last_error", "lfgpu_set_stream", "lfgpu_own_stream", "lfgpu_set_rng_exact_calls", "lfgpu_sync", "lfgpu_malloc",
    "lfgpu_free", "lfgpu_memcpy_h2d", "lfgpu_memcpy_d2h", "lfgpu_fp128_fft", "lfgpu_f64_2_fft", "lfgpu_gf2128_lch14_fft",
    "lfgpu_gf2128_rs_encode_rows", "lfgpu_gf2128_rs_encode_tableau", "lfgpu_fp128_rs_encode_rows", "lfgpu_fp256_rs_encode_rows", "lfgpu_column_commit", "lfgpu_column_leaves", "lfgpu_merkle_build_tree",
    "lfgpu_merkle_open", "lfgpu_sumcheck_partials", "lfgpu_qw_scatter", "lfgpu_dense_bind", "lfgpu_hquad_bind_h",
    "lfgpu_rows_axpy", "lfgpu_gather_columns", "lfgpu_field_binop", "lfgpu_fp128_fft_host", "lfgpu_f64_2_fft_host", "lfgpu_gf2128_lch14_fft_host",
    "lfgpu_gf2128_rs_encode_rows_host", "lfgpu_fp128_rs_encode_rows_host", "lfgpu_fp256_rs_encode_rows_host", "lfgpu_column_commit_host",
    "lfgpu_ligero_param_init", "lfgpu_ligero_commit", "lfgpu_ligero_commit_sharded", "lfgpu_ligero_row_shard", "lfgpu_ligero_layout_rows_sharded", "lfgpu_comm_selftest", "lfgpu_ligero_layout_rows", "lfgpu_ligero_encode_rows", "lfgpu_ligero_prover_from_slab", "lfgpu_ligero_low_degree_proof", "lfgpu_ligero_dot_proof",
    "lfgpu_ligero_inner_product_rows", "lfgpu_ligero_dot_proof_sparse",
    "lfgpu_ligero_quadratic_proof", "lfgpu_ligero_open", "lfgpu_ligero_tableau", "lfgpu_ligero_free",
    "lfgpu_quad_upload", "lfgpu_quad_free", "lfgpu_eval_quad", "lfgpu_quad_bind_g", "lfgpu_sumcheck_layer", "lfgpu_raw_eq2", "lfgpu_quad_bind_gh_all",
    # include/lfgpu_zk.h
    "lfgpu_transcript_new", "lfgpu_transcript_free", "lfgpu_transcript_get_ops", "lfgpu_transcript_write_bytes",
    "lfgpu_transcript_write_elt", "lfgpu_transcript_write_elt_array", "lfgpu_transcript_bytes", "lfgpu_transcript_write_elt_sized", "lfgpu_transcript_write_elt_array_sized", "lfgpu_sha256",
    "lfgpu_aes256_ecb_block", "lfgpu_host_gf2128_mul", "lfgpu_crypto_hw", "lfgpu_circuit_from_lfc1", "lfgpu_circuit_share", "lfgpu_circuit_get_info", "lfgpu_circuit_layer_info",
    "lfgpu_circuit_free", "lfgpu_zk_prover_new", "lfgpu_zk_prover_set_comm", "lfgpu_zk_prover_param", "lfgpu_zk_commit", "lfgpu_zk_prove",
    "lfgpu_zk_proof_write", "lfgpu_zk_timings", "lfgpu_zk_prover_free", "lfgpu_zk_verify", "lfgpu_zk_verify_committed",
]


class LigeroParam(C.Structure):
    """lfgpu_ligero_param == LigeroParam (reference lib/ligero/ligero_param.h:117-307)"""
    _fields_ = [(n, C.c_size_t) for n in (
        "nw", "nq", "rateinv", "nreq", "block_enc", "block", "dblock", "block_ext", "r", "w", "nwrow", "nqtriples",
        "nwqrow", "nrow", "mc_pathlen", "ildt", "idot", "iquad", "iw", "iq")]


RNG_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)  # RandomEngine::bytes
# round callback of lfgpu_sumcheck_layer: (user, hand, round, evals[3][2], challenge_out[2])
SC_ROUND_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))


class TranscriptOps(C.Structure):
    """lfgpu_transcript_ops (include/lfgpu_zk.h): the caller's Fiat-Shamir transcript behind function pointers"""
    _fields_ = [("user", C.c_void_p), ("write_bytes", C.c_void_p), ("write_elt", C.c_void_p),
                ("write_elt_array", C.c_void_p), ("gen_bytes", C.c_void_p), ("clone", C.c_void_p), ("free_clone", C.c_void_p),
                ("write_elt_sized", C.c_void_p), ("write_elt_array_sized", C.c_void_p)]  # 32-byte elements (Fp256Base)


class CircuitInfo(C.Structure):
    """lfgpu_circuit_info"""
    _fields_ = [("field", C.c_int)] + [(n, C.c_size_t) for n in (
        "nv", "nc", "npub_in", "subfield_boundary", "ninputs", "nl", "logv", "nterms")] + [("id", C.c_uint8 * 32)]


class LfGpuError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen liblfgpu.so and declare the ABI.  Raises (never falls back) when missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LfGpuError("%s not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (libamdhip64 under torch/lib).  When torch shares the process -- device
    # tensors, streams, torch.distributed -- that runtime must be the one already loaded when liblfgpu.so is opened;
    # loading the system runtime first and torch's afterwards leaves the process with two runtimes and lfgpu_init
    # fails.  Stand-alone use (examples/zk_flatsha.cc) needs no torch and binds the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u64, ci = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int
    pu64 = C.POINTER(C.c_uint64)
    sig = {
        "lfgpu_init": [ci, C.POINTER(vp)], "lfgpu_shutdown": [vp], "lfgpu_set_stream": [vp, vp], "lfgpu_own_stream": [vp], "lfgpu_set_rng_exact_calls": [vp, ci], "lfgpu_sync": [vp],
        "lfgpu_malloc": [vp, sz, C.POINTER(vp)], "lfgpu_free": [vp, vp],
        "lfgpu_memcpy_h2d": [vp, vp, vp, sz], "lfgpu_memcpy_d2h": [vp, vp, vp, sz],
        "lfgpu_fp128_fft": [vp, ci, sz, sz, pu64, u64, vp, sz],
        "lfgpu_f64_2_fft": [vp, ci, sz, sz, pu64, u64, vp, sz],
        "lfgpu_gf2128_lch14_fft": [vp, ci, ci, sz, C.c_uint, u64, vp, sz],
        "lfgpu_gf2128_rs_encode_rows": [vp, ci, sz, sz, sz, vp, sz],
        "lfgpu_gf2128_rs_encode_tableau": [vp, ci, sz, sz, sz, sz, sz, sz, vp, sz],
        "lfgpu_fp128_rs_encode_rows": [vp, sz, sz, sz, pu64, u64, vp, sz],
        "lfgpu_fp256_rs_encode_rows": [vp, sz, sz, sz, vp, sz],
        "lfgpu_column_commit": [vp, ci, sz, sz, sz, sz, vp, vp, vp, vp],
        "lfgpu_column_leaves": [vp, ci, sz, sz, sz, sz, vp, vp, vp],
        "lfgpu_merkle_build_tree": [vp, sz, vp, vp],
        "lfgpu_merkle_open": [vp, sz, vp, vp, sz, vp, sz, C.POINTER(sz)],
        "lfgpu_sumcheck_partials": [vp, ci, sz, vp, vp, pu64, pu64],
        "lfgpu_qw_scatter": [vp, ci, sz, vp, vp, ci, vp, sz, vp],
        "lfgpu_dense_bind": [vp, ci, sz, pu64, vp, vp],
        "lfgpu_hquad_bind_h": [vp, ci, sz, vp, vp, pu64, ci, vp, vp, C.POINTER(sz)],
        "lfgpu_rows_axpy": [vp, ci, sz, sz, vp, vp, vp, sz],
        "lfgpu_field_binop": [vp, ci, ci, sz, vp, vp, vp],
        "lfgpu_gather_columns": [vp, sz, sz, sz, vp, vp, sz, vp],
        "lfgpu_fp128_fft_host": [vp, ci, sz, pu64, u64, vp],
        "lfgpu_f64_2_fft_host": [vp, ci, sz, pu64, u64, vp],
        "lfgpu_gf2128_lch14_fft_host": [vp, ci, ci, C.c_uint, u64, vp],
        "lfgpu_gf2128_rs_encode_rows_host": [vp, ci, sz, sz, sz, vp, sz],
        "lfgpu_fp128_rs_encode_rows_host": [vp, sz, sz, sz, pu64, u64, vp, sz],
        "lfgpu_fp256_rs_encode_rows_host": [vp, sz, sz, sz, vp, sz],
        "lfgpu_column_commit_host": [vp, ci, sz, sz, sz, sz, vp, vp, vp, vp],
        "lfgpu_ligero_param_init": [C.POINTER(LigeroParam), ci, ci, sz, sz, sz, sz, sz],
        "lfgpu_ligero_commit": [vp, ci, ci, C.POINTER(LigeroParam), vp, sz, vp, RNG_FN, vp, vp, C.POINTER(vp)],
        "lfgpu_ligero_layout_rows": [ci, ci, C.POINTER(LigeroParam), vp, sz, vp, RNG_FN, vp, sz, sz, vp, vp],
        "lfgpu_ligero_commit_sharded": [vp, ci, ci, C.POINTER(LigeroParam), vp, sz, vp, RNG_FN, vp, vp, vp, C.POINTER(vp)],
        "lfgpu_ligero_row_shard": [C.POINTER(LigeroParam), ci, ci, C.POINTER(sz), C.POINTER(sz)],
        "lfgpu_ligero_layout_rows_sharded": [ci, ci, C.POINTER(LigeroParam), vp, sz, vp, RNG_FN, vp, vp, vp, vp],
        "lfgpu_comm_selftest": [vp],
        "lfgpu_zk_prover_set_comm": [vp, vp, sz],
        "lfgpu_ligero_encode_rows": [vp, ci, ci, C.POINTER(LigeroParam), sz, sz, vp, vp],
        "lfgpu_ligero_prover_from_slab": [vp, ci, ci, C.POINTER(LigeroParam), sz, sz, vp, vp, vp, C.POINTER(vp)],
        "lfgpu_ligero_low_degree_proof": [vp, vp, vp],
        "lfgpu_ligero_dot_proof": [vp, vp, vp],
        "lfgpu_ligero_inner_product_rows": [vp, ci, sz, sz, sz, sz, vp, sz, vp, vp, vp, sz, vp],
        "lfgpu_ligero_dot_proof_sparse": [vp, vp, sz, vp, vp, vp, sz, vp],
        "lfgpu_ligero_quadratic_proof": [vp, vp, vp, vp],
        "lfgpu_ligero_open": [vp, vp, vp, vp, vp, sz, C.POINTER(sz)],
        "lfgpu_ligero_tableau": [vp, C.POINTER(vp)],
        "lfgpu_ligero_free": [vp],
        "lfgpu_quad_upload": [vp, ci, sz, vp, vp, vp, vp, sz, vp, sz, C.POINTER(vp)],
        "lfgpu_quad_free": [vp],
        "lfgpu_eval_quad": [vp, sz, vp, vp, C.POINTER(ci)],
        "lfgpu_quad_bind_g": [vp, sz, vp, vp, pu64, pu64, vp, vp, C.POINTER(sz)],
        "lfgpu_sumcheck_layer": [vp, sz, vp, vp, pu64, pu64, sz, sz, vp, pu64, SC_ROUND_FN, vp, pu64, pu64, pu64],
        "lfgpu_raw_eq2": [vp, ci, sz, sz, vp, vp, pu64, vp],
        "lfgpu_quad_bind_gh_all": [vp, sz, vp, vp, pu64, pu64, sz, sz, vp, vp, pu64],
        "lfgpu_circuit_from_lfc1": [vp, vp, sz, C.POINTER(vp)],
        "lfgpu_circuit_share": [vp, vp, C.POINTER(vp)],
        "lfgpu_circuit_get_info": [vp, C.POINTER(CircuitInfo)],
        "lfgpu_circuit_layer_info": [vp, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)],
        "lfgpu_circuit_free": [vp],
        "lfgpu_zk_prover_new": [vp, vp, sz, sz, sz, C.POINTER(vp)],
        "lfgpu_zk_prover_param": [vp, C.POINTER(LigeroParam)],
        "lfgpu_zk_commit": [vp, vp, RNG_FN, vp, C.POINTER(TranscriptOps), vp],
        "lfgpu_zk_prove": [vp, vp, C.POINTER(TranscriptOps), C.POINTER(ci)],
        "lfgpu_zk_proof_write": [vp, vp, sz, C.POINTER(sz)],
        "lfgpu_zk_timings": [vp, C.POINTER(C.c_double)],
        "lfgpu_zk_prover_free": [vp],
        "lfgpu_zk_verify": [vp, vp, sz, sz, sz, vp, sz, vp, C.POINTER(TranscriptOps), C.POINTER(ci), C.POINTER(C.c_char_p)],
        "lfgpu_zk_verify_committed": [vp, vp, sz, sz, sz, vp, sz, vp, C.POINTER(TranscriptOps), C.POINTER(ci), C.POINTER(C.c_char_p)],
        "lfgpu_crypto_hw": [ci],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = ci, args
    L.lfgpu_last_error.restype, L.lfgpu_last_error.argtypes = C.c_char_p, [vp]
    L.lfgpu_transcript_new.restype, L.lfgpu_transcript_new.argtypes = vp, [vp, sz]
    for name, args in (("lfgpu_transcript_free", [vp]), ("lfgpu_transcript_get_ops", [vp, C.POINTER(TranscriptOps)]),
                       ("lfgpu_transcript_write_bytes", [vp, vp, sz]), ("lfgpu_transcript_write_elt", [vp, vp]),
                       ("lfgpu_transcript_write_elt_array", [vp, vp, sz]), ("lfgpu_transcript_bytes", [vp, vp, sz]),
                       ("lfgpu_transcript_write_elt_sized", [vp, vp, sz]), ("lfgpu_transcript_write_elt_array_sized", [vp, vp, sz, sz]),
                       ("lfgpu_sha256", [vp, sz, vp]), ("lfgpu_aes256_ecb_block", [vp, vp, vp]),
                       ("lfgpu_host_gf2128_mul", [pu64, pu64, pu64])):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = None, args
    _lib = L
    return L


def _u64x2(v):
    """int (canonical 128-bit image) or sequence of two u64 -> (c_uint64 * 2)"""
    if isinstance(v, int):
        v = (v & (2**64 - 1), v >> 64)
    return (C.c_uint64 * 2)(int(v[0]), int(v[1]))


def fp128_to_montgomery(x):
    return (x * (1 << 128)) % FP128_P


def fp128_from_montgomery(x):
    return (x * pow(1 << 128, -1, FP128_P)) % FP128_P


def f64_to_montgomery(x):
    return (x * (1 << 64)) % F64_P


def f64_from_montgomery(x):
    return (x * pow(1 << 64, -1, F64_P)) % F64_P


def _f64_2_omega(omega):
    """None = the reference's root of order 2^32 (real); else (re, im) Montgomery images"""
    if omega is None:
        omega = (f64_to_montgomery(F64_OMEGA32), 0)
    return (C.c_uint64 * 2)(int(omega[0]), int(omega[1]))


class LfGpu:
    """One context per process / GPU (lfgpu_ctx).  Pointer arguments are raw device
    addresses (e.g. torch.Tensor.data_ptr()); strides are in 16-byte elements."""

    def __init__(self, device=0, stream=None):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.lfgpu_init(int(device), C.byref(h))
        if rc != 0:
            raise LfGpuError("lfgpu_init(device=%d) failed with code %d (no usable MI355X?)" % (device, rc))
        self.h = h
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None):
            self.L.lfgpu_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise LfGpuError("lfgpu error %d: %s" % (rc, self.L.lfgpu_last_error(self.h).decode()))

    def set_stream(self, stream):
        self._ck(self.L.lfgpu_set_stream(self.h, C.c_void_p(int(stream))))

    def own_stream(self):
        """a non-blocking stream of the context's own: K contexts in K host threads then run concurrently on one device"""
        self._ck(self.L.lfgpu_own_stream(self.h))
        return self

    def set_rng_exact_calls(self, exact=True):
        """one RandomEngine call per element / nonce, as the reference draws (for engines that are not byte streams)"""
        self._ck(self.L.lfgpu_set_rng_exact_calls(self.h, 1 if exact else 0))

    def sync(self):
        self._ck(self.L.lfgpu_sync(self.h))

    # --- K1 FFT<Fp128>::fftb / fftf (reference lib/algebra/fft.h:185-201)
    def fp128_fft(self, d_ptr, rows, n, ld=None, forward=False, omega=None, omega_order=1 << 32):
        omega = _u64x2(fp128_to_montgomery(FP128_OMEGA32) if omega is None else omega)
        self._ck(self.L.lfgpu_fp128_fft(self.h, 1 if forward else 0, rows, n, omega, omega_order, C.c_void_p(d_ptr),
                                        n if ld is None else ld))

    # --- K1 over F64_2 = Fp2<Fp<1>> (reference lib/algebra/fft_test.cc:205-229); omega = (re, im) in Montgomery form
    def f64_2_fft(self, d_ptr, rows, n, ld=None, forward=False, omega=None, omega_order=1 << 32):
        omega = _f64_2_omega(omega)
        self._ck(self.L.lfgpu_f64_2_fft(self.h, 1 if forward else 0, rows, n, omega, omega_order, C.c_void_p(d_ptr),
                                        n if ld is None else ld))

    # --- K2 LCH14::FFT / IFFT (reference lib/gf2k/lch14.h:106-144)
    def gf2128_lch14_fft(self, d_ptr, rows, l, coset=0, ld=None, inverse=False, subfield_log_bits=4):
        self._ck(self.L.lfgpu_gf2128_lch14_fft(self.h, subfield_log_bits, 1 if inverse else 0, rows, l, coset,
                                               C.c_void_p(d_ptr), (1 << l) if ld is None else ld))

    # --- K3 LCH14ReedSolomon::interpolate over rows (reference lib/gf2k/lch14_reed_solomon.h:49-103)
    def gf2128_rs_encode_rows(self, d_ptr, nrow, n, m, ld=None, subfield_log_bits=4):
        self._ck(self.L.lfgpu_gf2128_rs_encode_rows(self.h, subfield_log_bits, nrow, n, m, C.c_void_p(d_ptr),
                                                    m if ld is None else ld))

    def gf2128_rs_encode_tableau(self, d_ptr, nrow, n1, n2, lo2, hi2, m, ld=None, subfield_log_bits=4):
        """all rows of a Ligero tableau in one launch: rows [lo2, hi2) have n2 valid values, the others n1"""
        self._ck(self.L.lfgpu_gf2128_rs_encode_tableau(self.h, subfield_log_bits, nrow, n1, n2, lo2, hi2, m, C.c_void_p(d_ptr),
                                                       m if ld is None else ld))

    # --- K4 ReedSolomon::interpolate over rows (reference lib/algebra/reed_solomon.h:93-110)
    def fp128_rs_encode_rows(self, d_ptr, nrow, n, m, ld=None, omega=None, omega_order=1 << 32):
        omega = _u64x2(fp128_to_montgomery(FP128_OMEGA32) if omega is None else omega)
        self._ck(self.L.lfgpu_fp128_rs_encode_rows(self.h, nrow, n, m, omega, omega_order, C.c_void_p(d_ptr),
                                                   m if ld is None else ld))

    # --- ReedSolomon<Fp256Base, FFTExtConvolution>::interpolate over rows of 32-byte elements (BASELINE config 5)
    def fp256_rs_encode_rows(self, d_ptr, nrow, n, m, ld=None):
        self._ck(self.L.lfgpu_fp256_rs_encode_rows(self.h, nrow, n, m, C.c_void_p(d_ptr), m if ld is None else ld))

    # --- inner_product_vector + layout_Aext on the device (reference lib/ligero/ligero_param.h:382-430)
    def ligero_inner_product_rows(self, field, w, r, ld, nrows, d_dense, ndense, scale, idx, val, d_rows):
        """rows[i][r + j] = scale * dense[i*w + j] (flat positions < ndense), then rows[pos(idx)] += val; the first
        r + w columns of all nrows rows are cleared first.  idx: strictly increasing flat indices (host)."""
        import numpy as np
        sc = np.ascontiguousarray(scale, dtype=np.uint64)
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        val = np.ascontiguousarray(val, dtype=np.uint64)
        self._ck(self.L.lfgpu_ligero_inner_product_rows(self.h, field, w, r, ld, nrows, C.c_void_p(d_dense), ndense,
                                                        C.c_void_p(sc.ctypes.data), C.c_void_p(idx.ctypes.data),
                                                        C.c_void_p(val.ctypes.data), len(idx), C.c_void_p(d_rows)))

    # --- K5+K6 MerkleCommitment::commit (reference lib/merkle/merkle_commitment.h:50-64)
    def column_commit(self, field, nrow, ld, col0, ncols, d_T, d_nonces, d_layers):
        root = (C.c_uint8 * 32)()
        self._ck(self.L.lfgpu_column_commit(self.h, field, nrow, ld, col0, ncols, C.c_void_p(d_T),
                                            C.c_void_p(d_nonces), C.c_void_p(d_layers), root))
        return bytes(root)

    def merkle_build_tree(self, n, d_layers):
        root = (C.c_uint8 * 32)()
        self._ck(self.L.lfgpu_merkle_build_tree(self.h, n, C.c_void_p(d_layers), root))
        return bytes(root)

    def merkle_open(self, n, d_layers, pos):
        pos_a = (C.c_size_t * len(pos))(*pos)
        cap = max(1, len(pos) * 64)
        buf = (C.c_uint8 * (32 * cap))()
        npath = C.c_size_t()
        self._ck(self.L.lfgpu_merkle_open(self.h, n, C.c_void_p(d_layers), pos_a, len(pos), buf, cap, C.byref(npath)))
        raw = bytes(buf)
        return [raw[32 * i:32 * i + 32] for i in range(npath.value)]

    # --- K7 loop of ProverLayers::evaluations (reference lib/sumcheck/prover_layers.h:365-388)
    def sumcheck_partials(self, field, n, d_QW, d_W):
        a0, a2 = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        self._ck(self.L.lfgpu_sumcheck_partials(self.h, field, n, C.c_void_p(d_QW), C.c_void_p(d_W), a0, a2))
        return (a0[0], a0[1]), (a2[0], a2[1])

    # --- K8 QW scatter (reference lib/sumcheck/prover_layers.h:239-243)
    def qw_scatter(self, field, n, d_hc, d_vc, hand, d_Wother, nqw, d_QW):
        self._ck(self.L.lfgpu_qw_scatter(self.h, field, n, C.c_void_p(d_hc), C.c_void_p(d_vc), hand,
                                         C.c_void_p(d_Wother), nqw, C.c_void_p(d_QW)))

    # --- K9 Dense::bind / HQuad::bind_h (reference lib/arrays/dense.h:70-87, lib/sumcheck/hquad.h:90-123)
    def dense_bind(self, field, n0, r, d_in, d_out):
        self._ck(self.L.lfgpu_dense_bind(self.h, field, n0, _u64x2(r), C.c_void_p(d_in), C.c_void_p(d_out)))
        return (n0 + 1) // 2

    def hquad_bind_h(self, field, n, d_hc, d_vc, r, hand, d_hc_out, d_vc_out):
        n_out = C.c_size_t()
        self._ck(self.L.lfgpu_hquad_bind_h(self.h, field, n, C.c_void_p(d_hc), C.c_void_p(d_vc), _u64x2(r), hand,
                                           C.c_void_p(d_hc_out), C.c_void_p(d_vc_out), C.byref(n_out)))
        return n_out.value

    # --- Eqs::raw_eq2 (reference lib/arrays/eqs.h:46-80)
    def raw_eq2(self, field, logn, n, G0, G1, alpha, d_eq):
        """eq[i] = EQ(G0, i) + alpha EQ(G1, i), i < n; G0 / G1: numpy uint64[logn][2] host arrays"""
        import numpy as np
        G0, G1 = np.ascontiguousarray(G0, dtype=np.uint64), np.ascontiguousarray(G1, dtype=np.uint64)
        self._ck(self.L.lfgpu_raw_eq2(self.h, field, logn, n, C.c_void_p(G0.ctypes.data), C.c_void_p(G1.ctypes.data),
                                      _u64x2(alpha), C.c_void_p(d_eq)))

    def field_binop(self, field, op, n, d_a, d_b, d_out):
        """element-wise Field::addf/subf/mulf (op 0/1/2)"""
        self._ck(self.L.lfgpu_field_binop(self.h, field, op, n, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_out)))

    # --- K12 row combinations (reference lib/ligero/ligero_prover.h:281-291,346-351)
    def rows_axpy(self, field, nrows, n, d_y, u_host, d_T, ld):
        self._ck(self.L.lfgpu_rows_axpy(self.h, field, nrows, n, C.c_void_p(d_y), C.c_void_p(u_host.ctypes.data),
                                        C.c_void_p(d_T), ld))

    def gather_columns(self, nrow, ld, col0, d_T, idx, d_req):
        idx_a = (C.c_size_t * len(idx))(*idx)
        self._ck(self.L.lfgpu_gather_columns(self.h, nrow, ld, col0, C.c_void_p(d_T), idx_a, len(idx),
                                             C.c_void_p(d_req)))

    # --- host-buffer conveniences (numpy uint64[...,2] arrays, modified in place)
    def fp128_fft_host(self, a, forward=False, omega=None, omega_order=1 << 32):
        omega = _u64x2(fp128_to_montgomery(FP128_OMEGA32) if omega is None else omega)
        self._ck(self.L.lfgpu_fp128_fft_host(self.h, 1 if forward else 0, a.shape[0], omega, omega_order,
                                             C.c_void_p(a.ctypes.data)))

    def f64_2_fft_host(self, a, forward=False, omega=None, omega_order=1 << 32):
        self._ck(self.L.lfgpu_f64_2_fft_host(self.h, 1 if forward else 0, a.shape[0], _f64_2_omega(omega), omega_order,
                                             C.c_void_p(a.ctypes.data)))

    def gf2128_lch14_fft_host(self, a, l, coset=0, inverse=False, subfield_log_bits=4):
        self._ck(self.L.lfgpu_gf2128_lch14_fft_host(self.h, subfield_log_bits, 1 if inverse else 0, l, coset,
                                                    C.c_void_p(a.ctypes.data)))

    def gf2128_rs_encode_rows_host(self, T, nrow, n, m, ld, subfield_log_bits=4):
        self._ck(self.L.lfgpu_gf2128_rs_encode_rows_host(self.h, subfield_log_bits, nrow, n, m,
                                                         C.c_void_p(T.ctypes.data), ld))

    def column_commit_host(self, field, T, nrow, ld, col0, ncols, nonces, layers=None):
        root = (C.c_uint8 * 32)()
        self._ck(self.L.lfgpu_column_commit_host(self.h, field, nrow, ld, col0, ncols, C.c_void_p(T.ctypes.data),
                                                 C.c_void_p(nonces.ctypes.data),
                                                 C.c_void_p(layers.ctypes.data) if layers is not None else None, root))
        return bytes(root)


def ligero_param(field, nw, nq, rateinv, nreq, block_enc, subfield_log_bits=4):
    """LigeroParam(nw, nq, rateinv, nreq, block_enc) (reference lib/ligero/ligero_param.h:172-178)"""
    p = LigeroParam()
    rc = load_library().lfgpu_ligero_param_init(C.byref(p), field, subfield_log_bits, nw, nq, rateinv, nreq, block_enc)
    if rc != 0:
        raise LfGpuError("LigeroParam layout failed (block_enc too large / too small)")
    return p


class LigeroProver:
    """Mirror of LigeroProver<Field, InterpolatorFactory> (reference lib/ligero/ligero_prover.h:34-359)
    over the C ABI: the tableau lives in HBM; the caller owns the transcript and the RandomEngine.
    `rng_bytes(n) -> bytes` plays RandomEngine::bytes."""

    def __init__(self, gpu, field, param, subfield_log_bits=4):
        self.gpu, self.field, self.p, self.k = gpu, field, param, subfield_log_bits
        self.h = None

    def commit(self, W, subfield_boundary, lqc, rng_bytes):
        """returns the 32-byte commitment root (LigeroProver::commit, :58-79, without ts.write)"""
        import numpy as np
        gpu = self.gpu

        def cb(_user, buf, n):
            data = rng_bytes(n)
            C.memmove(buf, data, n)

        self._cb = RNG_FN(cb)
        lq = np.ascontiguousarray(np.asarray(lqc, dtype=np.uint64).reshape(-1))
        W = np.ascontiguousarray(W)
        root = (C.c_uint8 * 32)()
        h = C.c_void_p()
        gpu._ck(gpu.L.lfgpu_ligero_commit(gpu.h, self.field, self.k, C.byref(self.p), C.c_void_p(W.ctypes.data),
                                           subfield_boundary, C.c_void_p(lq.ctypes.data) if lq.size else None,
                                           self._cb, None, root, C.byref(h)))
        self.h = h
        return bytes(root)

    def low_degree_proof(self, u_ldt):
        import numpy as np
        y = np.zeros((self.p.block, 2), dtype=np.uint64)
        u = np.ascontiguousarray(u_ldt)
        self.gpu._ck(self.gpu.L.lfgpu_ligero_low_degree_proof(self.h, C.c_void_p(u.ctypes.data), C.c_void_p(y.ctypes.data)))
        return y

    def dot_proof(self, A):
        import numpy as np
        y = np.zeros((self.p.dblock, 2), dtype=np.uint64)
        A = np.ascontiguousarray(A)
        self.gpu._ck(self.gpu.L.lfgpu_ligero_dot_proof(self.h, C.c_void_p(A.ctypes.data), C.c_void_p(y.ctypes.data)))
        return y

    def dot_proof_sparse(self, d_dense, ndense, scale, idx, val):
        """dot_proof with A built on the device: A[t] = scale * d_dense[t] (device pointer), A[idx] += val."""
        import numpy as np
        y = np.zeros((self.p.dblock, 2), dtype=np.uint64)
        sc = np.ascontiguousarray(scale, dtype=np.uint64)
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        val = np.ascontiguousarray(val, dtype=np.uint64)
        self.gpu._ck(self.gpu.L.lfgpu_ligero_dot_proof_sparse(self.h, C.c_void_p(d_dense), ndense, C.c_void_p(sc.ctypes.data),
                                                              C.c_void_p(idx.ctypes.data), C.c_void_p(val.ctypes.data), len(idx),
                                                              C.c_void_p(y.ctypes.data)))
        return y

    def quadratic_proof(self, u_quad):
        import numpy as np
        y0 = np.zeros((self.p.r, 2), dtype=np.uint64)
        y2 = np.zeros((self.p.dblock - self.p.block, 2), dtype=np.uint64)
        u = np.ascontiguousarray(u_quad)
        self.gpu._ck(self.gpu.L.lfgpu_ligero_quadratic_proof(self.h, C.c_void_p(u.ctypes.data) if u.size else None,
                                                             C.c_void_p(y0.ctypes.data), C.c_void_p(y2.ctypes.data)))
        return y0, y2

    def open(self, idx, rows=None):
        """compute_req + MerkleCommitment::open -> (req[nrow][nreq], nonces[nreq], path digests); a slab prover
        (parallel.GpuEngine.slab_prover) returns its own `rows` rows of req"""
        import numpy as np
        p = self.p
        idx_a = (C.c_size_t * p.nreq)(*idx)
        req = np.zeros((p.nrow if rows is None else rows, p.nreq, 2), dtype=np.uint64)
        nonces = np.zeros((p.nreq, 32), dtype=np.uint8)
        cap = p.nreq * p.mc_pathlen + 1
        path = np.zeros((cap, 32), dtype=np.uint8)
        npath = C.c_size_t()
        self.gpu._ck(self.gpu.L.lfgpu_ligero_open(self.h, idx_a, C.c_void_p(req.ctypes.data), C.c_void_p(nonces.ctypes.data),
                                                  C.c_void_p(path.ctypes.data), cap, C.byref(npath)))
        return req, nonces, [bytes(path[i]) for i in range(npath.value)]

    def tableau_ptr(self):
        d = C.c_void_p()
        self.gpu._ck(self.gpu.L.lfgpu_ligero_tableau(self.h, C.byref(d)))
        return d.value

    def close(self):
        if self.h:
            self.gpu.L.lfgpu_ligero_free(self.h)
            self.h = None


class Quad:
    """One sumcheck layer resident on the device (mirror of Quad<Field>, reference
    lib/sumcheck/quad.h:55-226): eval (ProverLayers::eval_quad) and bind_g."""

    def __init__(self, gpu, field, g, h0, h1, vi, kvec, nv):
        import numpy as np
        self.gpu, self.field, self.n, self.nv = gpu, field, len(g), nv
        arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (g, h0, h1, vi)]
        kvec = np.ascontiguousarray(kvec)
        h = C.c_void_p()
        gpu._ck(gpu.L.lfgpu_quad_upload(gpu.h, field, self.n, *[C.c_void_p(a.ctypes.data) for a in arrs], len(kvec),
                                        C.c_void_p(kvec.ctypes.data), nv, C.byref(h)))
        self.h = h

    def eval(self, nw, d_W, d_V):
        ok = C.c_int()
        self.gpu._ck(self.gpu.L.lfgpu_eval_quad(self.h, nw, C.c_void_p(d_W), C.c_void_p(d_V), C.byref(ok)))
        return bool(ok.value)

    def bind_g(self, logv, G0, G1, alpha, beta, d_hc_out, d_vc_out):
        import numpy as np
        G0, G1 = np.ascontiguousarray(G0), np.ascontiguousarray(G1)
        n_out = C.c_size_t()
        self.gpu._ck(self.gpu.L.lfgpu_quad_bind_g(self.h, logv, C.c_void_p(G0.ctypes.data), C.c_void_p(G1.ctypes.data),
                                                  _u64x2(alpha), _u64x2(beta), C.c_void_p(d_hc_out), C.c_void_p(d_vc_out),
                                                  C.byref(n_out)))
        return n_out.value

    def bind_gh_all(self, logv, G0, G1, alpha, beta, logw, nw, H0, H1):
        """Quad::bind_gh_all (the verifier's combined bind) -> (lo, hi)"""
        import numpy as np
        G0, G1, H0, H1 = (np.ascontiguousarray(a) for a in (G0, G1, H0, H1))
        out = (C.c_uint64 * 2)()
        self.gpu._ck(self.gpu.L.lfgpu_quad_bind_gh_all(self.h, logv, C.c_void_p(G0.ctypes.data), C.c_void_p(G1.ctypes.data),
                                                       _u64x2(alpha), _u64x2(beta), logw, nw, C.c_void_p(H0.ctypes.data),
                                                       C.c_void_p(H1.ctypes.data), out))
        return (out[0], out[1])

    def sumcheck_layer(self, logv, G0, G1, alpha, beta, logw, nw, d_W, wc_in, round_cb):
        """ProverLayers::layer (logc = 0) incl. bind_g; round_cb(hand, round, evals[3]) -> challenge, evals and
        challenge as (lo, hi) pairs.  Returns (wc_out[2], challenges[2][logw], bound_quad)."""
        import numpy as np
        G0, G1 = np.ascontiguousarray(G0), np.ascontiguousarray(G1)

        def cb(_user, hand, rnd, evals, out):
            ev = [(evals[2 * k], evals[2 * k + 1]) for k in range(3)]
            r = round_cb(hand, rnd, ev)
            out[0], out[1] = int(r[0]), int(r[1])

        cfn = SC_ROUND_FN(cb)
        wci = (C.c_uint64 * 4)(int(wc_in[0][0]), int(wc_in[0][1]), int(wc_in[1][0]), int(wc_in[1][1]))
        wco = (C.c_uint64 * 4)()
        gout = (C.c_uint64 * (4 * max(1, logw)))()
        bq = (C.c_uint64 * 2)()
        self.gpu._ck(self.gpu.L.lfgpu_sumcheck_layer(self.h, logv, C.c_void_p(G0.ctypes.data), C.c_void_p(G1.ctypes.data),
                                                     _u64x2(alpha), _u64x2(beta), logw, nw, C.c_void_p(d_W), wci, cfn, None,
                                                     wco, gout, bq))
        ch = [[(gout[(h * logw + r) * 2], gout[(h * logw + r) * 2 + 1]) for r in range(logw)] for h in range(2)]
        return [(wco[0], wco[1]), (wco[2], wco[3])], ch, (bq[0], bq[1])

    def close(self):
        if self.h:
            self.gpu.L.lfgpu_quad_free(self.h)
            self.h = None


class FsTranscript:
    """The library's built-in Fiat-Shamir transcript (reference lib/random/transcript.h:33-190); host only."""

    def __init__(self, init=b""):
        self.L = load_library()
        self.h = self.L.lfgpu_transcript_new(bytes(init), len(init))
        if not self.h:
            raise LfGpuError("lfgpu_transcript_new failed")

    def write_bytes(self, data):
        self.L.lfgpu_transcript_write_bytes(self.h, bytes(data), len(data))

    def write_elt(self, e16):
        self.L.lfgpu_transcript_write_elt(self.h, bytes(e16))

    def write_array(self, elts):
        b = b"".join(bytes(e) for e in elts)
        self.L.lfgpu_transcript_write_elt_array(self.h, b, len(elts))

    def write_elt_sized(self, e):
        self.L.lfgpu_transcript_write_elt_sized(self.h, bytes(e), len(e))

    def write_array_sized(self, elts, nbytes):
        b = b"".join(bytes(e) for e in elts)
        self.L.lfgpu_transcript_write_elt_array_sized(self.h, b, len(elts), nbytes)

    def bytes(self, n):
        buf = C.create_string_buffer(n)
        self.L.lfgpu_transcript_bytes(self.h, buf, n)
        return buf.raw

    def ops(self):
        o = TranscriptOps()
        self.L.lfgpu_transcript_get_ops(self.h, C.byref(o))
        return o

    def close(self):
        if self.h:
            self.L.lfgpu_transcript_free(self.h)
            self.h = None


def sha256(data):
    out = C.create_string_buffer(32)
    load_library().lfgpu_sha256(bytes(data), len(data), out)
    return out.raw


def aes256_ecb_block(key, block):
    out = C.create_string_buffer(16)
    load_library().lfgpu_aes256_ecb_block(bytes(key), bytes(block), out)
    return out.raw


class Circuit:
    """A circuit parsed from the reference's LFC1 wire bytes, layers resident on the device (lfgpu_circuit)."""

    def __init__(self, gpu, lfc1_bytes, _shared_from=None):
        self.gpu = gpu
        h = C.c_void_p()
        if _shared_from is not None:
            gpu._ck(gpu.L.lfgpu_circuit_share(gpu.h, _shared_from.h, C.byref(h)))
        else:
            raw = bytes(lfc1_bytes)
            gpu._ck(gpu.L.lfgpu_circuit_from_lfc1(gpu.h, raw, len(raw), C.byref(h)))
        self.h = h
        self.info = CircuitInfo()
        gpu._ck(gpu.L.lfgpu_circuit_get_info(self.h, C.byref(self.info)))

    def share(self, gpu):
        """a handle on the same device-resident circuit for another context of the same device (lfgpu_circuit_share)"""
        return Circuit(gpu, None, _shared_from=self)

    def layer(self, i):
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.gpu._ck(self.gpu.L.lfgpu_circuit_layer_info(self.h, i, C.byref(a), C.byref(b), C.byref(c)))
        return dict(logw=a.value, nw=b.value, nterms=c.value)

    def close(self):
        if self.h:
            self.gpu.L.lfgpu_circuit_free(self.h)
            self.h = None


class ZkProver:
    """Mirror of ZkProver<Field, RSFactory> (reference lib/zk/zk_prover.h:45-149) over include/lfgpu_zk.h: the whole
    host control flow runs in the library's C++; `rng_bytes(n) -> bytes` plays the RandomEngine."""

    def __init__(self, gpu, circuit, rate=7, nreq=132, block_enc=0):
        self.gpu, self.circuit = gpu, circuit
        h = C.c_void_p()
        gpu._ck(gpu.L.lfgpu_zk_prover_new(gpu.h, circuit.h, rate, nreq, block_enc, C.byref(h)))
        self.h = h
        self.param = LigeroParam()
        gpu._ck(gpu.L.lfgpu_zk_prover_param(self.h, C.byref(self.param)))

    def set_comm(self, comm, min_tableau_bytes=0):
        """lfgpu_zk_prover_set_comm: `comm` = parallel.TorchComm (or None); tableaux of at least min_tableau_bytes are
        committed with their rows sharded over the communicator's GPUs, smaller ones replicated"""
        self._comm = comm
        self.gpu._ck(self.gpu.L.lfgpu_zk_prover_set_comm(self.h, C.byref(comm.ops) if comm is not None else None, min_tableau_bytes))

    def commit(self, W, rng_bytes, transcript):
        import numpy as np

        def cb(_user, buf, n):
            C.memmove(buf, rng_bytes(n), n)

        self._cb = RNG_FN(cb)
        W = np.ascontiguousarray(W)
        root = (C.c_uint8 * 32)()
        ops = transcript.ops()
        self.gpu._ck(self.gpu.L.lfgpu_zk_commit(self.h, C.c_void_p(W.ctypes.data), self._cb, None, C.byref(ops), root))
        return bytes(root)

    def prove(self, W, transcript):
        """-> True (proof held by the object; wire() serializes it) or False (witness does not satisfy the circuit)"""
        import numpy as np
        W = np.ascontiguousarray(W)
        ok = C.c_int()
        ops = transcript.ops()
        self.gpu._ck(self.gpu.L.lfgpu_zk_prove(self.h, C.c_void_p(W.ctypes.data), C.byref(ops), C.byref(ok)))
        return bool(ok.value)

    def wire(self):
        """ZkProof::write bytes"""
        n = C.c_size_t()
        self.gpu._ck(self.gpu.L.lfgpu_zk_proof_write(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self.gpu._ck(self.gpu.L.lfgpu_zk_proof_write(self.h, buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    def timings(self):
        ms = (C.c_double * 6)()
        self.gpu._ck(self.gpu.L.lfgpu_zk_timings(self.h, ms))
        return dict(zip(("commit", "prove", "eval_circuit", "sumcheck", "constraints", "ligero_prove"), list(ms)))

    def close(self):
        if self.h:
            self.gpu.L.lfgpu_zk_prover_free(self.h)
            self.h = None


def zk_verify(gpu, circuit, proof, pub, transcript, rate=7, nreq=132, block_enc=0, committed=False):
    """ZkVerifier::recv_commitment + verify over the wire bytes -> (accepted, reason); committed=True: the caller has already
    written the commitment root to the transcript (ZkVerifier::recv_commitment as a separate step, as in run_mdoc_verifier)"""
    import numpy as np
    pub = np.ascontiguousarray(pub)
    ok, why = C.c_int(), C.c_char_p()
    ops = transcript.ops()
    raw = bytes(proof)
    fn = gpu.L.lfgpu_zk_verify_committed if committed else gpu.L.lfgpu_zk_verify
    gpu._ck(fn(gpu.h, circuit.h, rate, nreq, block_enc, raw, len(raw), C.c_void_p(pub.ctypes.data) if pub.size else None,
               C.byref(ops), C.byref(ok), C.byref(why)))
    return bool(ok.value), (why.value or b"").decode()
