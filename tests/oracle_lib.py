"""ctypes bindings for the CPU oracle (oracle/liblforacle.so) and, when it has
been built in this container, the compiled reference (oracle/_ref/liblfref.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GF, FP = 4, 6  # FieldID (lib/proto/circuit_io.h:24-36)


class Elt(C.Structure):
    _fields_ = [("l", C.c_uint64 * 2)]


class E32(C.Structure):
    """lfo_e32: Fp256Base::Elt image (Montgomery form, 4 x u64 LE)"""
    _fields_ = [("l", C.c_uint64 * 4)]


P256 = 1  # FieldID P256_ID
P256_P = 2**256 - 2**224 + 2**192 + 2**96 - 1


def e32(a):
    a = np.asarray(a, dtype=np.uint64).reshape(4)
    e = E32()
    for i in range(4):
        e.l[i] = int(a[i])
    return e


def arr32(e):
    return np.array([e.l[i] for i in range(4)], dtype=np.uint64)


class GfCtx(C.Structure):
    _fields_ = [("k", C.c_uint), ("sub_bits", C.c_uint), ("g", Elt), ("beta", Elt * 32),
                ("w_hat", (Elt * 32) * 32)]


def _build_oracle():
    so = os.path.join(ORACLE_DIR, "liblforacle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("lf_oracle.c", "lf_oracle_p256.c", "lf_oracle.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liblforacle.so"], stdout=subprocess.DEVNULL)
    return so


def P(a):
    """numpy uint64 array -> void*"""
    return a.ctypes.data_as(C.c_void_p)


def elt(a):
    a = np.asarray(a, dtype=np.uint64).reshape(2)
    e = Elt()
    e.l[0], e.l[1] = int(a[0]), int(a[1])
    return e


def arr(e):
    return np.array([e.l[0], e.l[1]], dtype=np.uint64)


_lib = None


def oracle():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(_build_oracle())
    sz, u64, vp, ci = C.c_size_t, C.c_uint64, C.c_void_p, C.c_int
    sig = {
        "lfo_p256_add": (E32, [E32, E32]), "lfo_p256_sub": (E32, [E32, E32]), "lfo_p256_mul": (E32, [E32, E32]),
        "lfo_p256_to_mont": (E32, [E32]), "lfo_p256_from_mont": (E32, [E32]), "lfo_p256_of_scalar": (E32, [u64]),
        "lfo_p256_inv": (E32, [E32]), "lfo_p256_to_bytes": (None, [vp, E32]), "lfo_p256_fill": (None, [u64, sz, vp]),
        "lfo_p256_omega": (None, [vp, vp]), "lfo_p256_r2hc": (None, [vp, sz]), "lfo_p256_hc2r": (None, [vp, sz]),
        "lfo_p256_rs_interpolate": (None, [sz, sz, vp]),
        "lfo_column_leaves32": (None, [sz, sz, sz, sz, vp, vp, vp]),
        "lfo_column_commit32": (None, [sz, sz, sz, sz, vp, vp, vp, vp]),
        "lfo_gf_mul": (Elt, [Elt, Elt]), "lfo_gf_mul_bitserial": (Elt, [Elt, Elt]), "lfo_gf_inv": (Elt, [Elt]),
        "lfo_gf_ctx_init": (None, [C.POINTER(GfCtx), C.c_uint]),
        "lfo_gf_of_scalar": (Elt, [C.POINTER(GfCtx), u64]),
        "lfo_gf_poly_evaluation_point": (Elt, [C.POINTER(GfCtx), C.c_uint]),
        "lfo_lch14_twiddle": (Elt, [C.POINTER(GfCtx), C.c_uint, u64]),
        "lfo_lch14_fft": (None, [C.POINTER(GfCtx), C.c_uint, u64, vp]),
        "lfo_lch14_ifft": (None, [C.POINTER(GfCtx), C.c_uint, u64, vp]),
        "lfo_lch14_bidirectional_fft": (None, [C.POINTER(GfCtx), C.c_uint, u64, vp]),
        "lfo_lch14_rs_interpolate": (None, [C.POINTER(GfCtx), sz, sz, vp]),
        "lfo_fp_add": (Elt, [Elt, Elt]), "lfo_fp_sub": (Elt, [Elt, Elt]), "lfo_fp_mul": (Elt, [Elt, Elt]),
        "lfo_fp_to_mont": (Elt, [Elt]), "lfo_fp_from_mont": (Elt, [Elt]), "lfo_fp_of_scalar": (Elt, [u64]),
        "lfo_fp_inv": (Elt, [Elt]), "lfo_fp_omega32": (Elt, []),
        "lfo_fp_fftb": (None, [vp, sz, Elt, u64]), "lfo_fp_fftf": (None, [vp, sz, Elt, u64]),
        "lfo_fp_rs_interpolate": (None, [sz, sz, vp]),
        "lfo_f64_add": (u64, [u64, u64]), "lfo_f64_sub": (u64, [u64, u64]), "lfo_f64_mul": (u64, [u64, u64]),
        "lfo_f64_of_scalar": (u64, [u64]), "lfo_f64_from_mont": (u64, [u64]), "lfo_f64_inv": (u64, [u64]),
        "lfo_f64_omega32": (u64, []),
        "lfo_f64_2_add": (Elt, [Elt, Elt]), "lfo_f64_2_sub": (Elt, [Elt, Elt]), "lfo_f64_2_mul": (Elt, [Elt, Elt]),
        "lfo_f64_2_inv": (Elt, [Elt]),
        "lfo_f64_2_fftb": (None, [vp, sz, Elt, u64]), "lfo_f64_2_fftf": (None, [vp, sz, Elt, u64]),
        "lfo_f64_2_bogorng_fill": (None, [u64, ci, sz, vp]),
        "lfo_add": (Elt, [ci, Elt, Elt]), "lfo_sub": (Elt, [ci, Elt, Elt]), "lfo_mul": (Elt, [ci, Elt, Elt]),
        "lfo_sha256_init": (None, [vp]), "lfo_sha256_update": (None, [vp, vp, sz]),
        "lfo_sha256_final": (None, [vp, vp]),
        "lfo_merkle_build_tree": (None, [sz, vp, vp]),
        "lfo_column_leaves": (None, [ci, sz, sz, sz, sz, vp, vp, vp]),
        "lfo_column_commit": (None, [ci, sz, sz, sz, sz, vp, vp, vp, vp]),
        "lfo_sumcheck_partials": (None, [ci, sz, vp, vp, C.POINTER(Elt), C.POINTER(Elt)]),
        "lfo_sumcheck_evaluations": (None, [ci, C.POINTER(GfCtx), sz, Elt, vp, vp, Elt, vp]),
        "lfo_dense_bind": (sz, [ci, sz, Elt, vp, vp]),
        "lfo_hquad_bind_h": (sz, [ci, sz, vp, vp, Elt, ci]),
        "lfo_qw_scatter": (None, [ci, sz, vp, vp, ci, vp, sz, vp]),
        "lfo_raw_eq2": (None, [ci, sz, sz, vp, vp, Elt, vp]),
        "lfo_eval_quad": (ci, [ci, sz, vp, vp, vp, vp, vp, sz, vp, vp]),
        "lfo_quad_bind_g": (sz, [ci, sz, vp, vp, vp, vp, vp, sz, vp, vp, Elt, Elt, vp, vp]),
        "lfo_quad_bind_gh_all": (Elt, [ci, sz, vp, vp, vp, vp, vp, sz, sz, vp, vp, Elt, Elt, sz, sz, vp, vp]),
        "lfo_axpy": (None, [ci, sz, vp, Elt, vp]), "lfo_vaxpy": (None, [ci, sz, vp, vp, vp]),
        "lfo_fp_bogorng_fill": (None, [u64, sz, vp]), "lfo_gf_fill": (None, [u64, sz, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


_ctx = {}


def gf_ctx(k):
    if k not in _ctx:
        c = GfCtx()
        oracle().lfo_gf_ctx_init(C.byref(c), k)
        _ctx[k] = c
    return _ctx[k]


_ref = None


def ref_path():
    return os.path.join(ORACLE_DIR, "_ref", "liblfref.so")


def have_ref():
    return os.path.exists(ref_path())


def ref():
    """The compiled REAL reference (oracle/_ref); None when not built."""
    global _ref
    if _ref is not None:
        return _ref
    if not have_ref():
        return None
    L = C.CDLL(ref_path())
    sz, u64, vp, ci = C.c_size_t, C.c_uint64, C.c_void_p, C.c_int
    sig = {
        "ref_gf_mul": (None, [vp, vp, vp]), "ref_gf_inv": (None, [vp, vp]),
        "ref_gf_of_scalar": (None, [ci, u64, vp]), "ref_gf_beta": (None, [ci, sz, vp]),
        "ref_gf_poly_evaluation_point": (None, [ci, sz, vp]),
        "ref_lch14_twiddle": (None, [ci, sz, sz, vp]),
        "ref_lch14_fft": (None, [ci, ci, sz, sz, vp]),
        "ref_lch14_rs_interpolate": (None, [ci, sz, sz, vp]),
        "ref_lch14_rs_encode_rows": (None, [ci, sz, sz, sz, vp, sz]),
        "ref_fp_mul": (None, [vp, vp, vp]), "ref_fp_add": (None, [vp, vp, vp]), "ref_fp_sub": (None, [vp, vp, vp]),
        "ref_fp_inv": (None, [vp, vp]), "ref_fp_of_scalar": (None, [u64, vp]), "ref_fp_from_mont": (None, [vp, vp]),
        "ref_fp_omega32": (None, [vp]), "ref_fp_bogorng_fill": (None, [u64, sz, vp]),
        "ref_fp_fft": (None, [ci, sz, vp]), "ref_fp_rs_interpolate": (None, [sz, sz, vp]),
        "ref_f64_2_binop": (None, [ci, vp, vp, vp]), "ref_f64_2_of_scalar": (None, [u64, u64, vp]),
        "ref_f64_2_omega32": (None, [vp]), "ref_f64_2_bogorng_fill": (None, [u64, ci, sz, vp]),
        "ref_f64_2_fft": (None, [ci, sz, vp, vp]),
        "ref_merkle_build_tree": (None, [sz, vp, vp]),
        "ref_column_commit": (None, [ci, sz, sz, sz, sz, vp, vp, vp]),
        "ref_sumcheck_evaluations": (None, [ci, sz, vp, vp, vp, vp, vp]),
        "ref_dense_bind": (sz, [ci, sz, vp, vp]),
        "ref_hquad_bind_h": (sz, [ci, sz, vp, vp, vp, ci]),
        "ref_eval_quad": (ci, [ci, sz, vp, vp, vp, vp, sz, vp, sz, sz, vp, vp]),
        "ref_quad_bind_g": (sz, [ci, sz, vp, vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, vp, vp]),
        "ref_raw_eq2": (None, [ci, sz, sz, vp, vp, vp, vp]),
        "ref_p256_mul": (None, [vp, vp, vp]), "ref_p256_add": (None, [vp, vp, vp]), "ref_p256_sub": (None, [vp, vp, vp]),
        "ref_p256_inv": (None, [vp, vp]), "ref_p256_of_scalar": (None, [u64, vp]), "ref_p256_to_bytes": (None, [vp, vp]),
        "ref_p256_omega": (None, [vp, vp]), "ref_p256_rfft": (None, [ci, sz, vp]),
        "ref_p256_rs_interpolate": (None, [sz, sz, vp]), "ref_p256_rs_encode_rows": (None, [sz, sz, sz, vp, sz]),
        "ref_p256_column_commit": (None, [sz, sz, sz, sz, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _ref = L
    return L


def rand_elts(rng, n, field=GF):
    """n random field elements as uint64[n,2]; Fp128 values are < p (valid Montgomery images)."""
    a = rng.integers(0, 2**64, size=(n, 2), dtype=np.uint64)
    if field == FP:
        # p = 2^128 - 2^108 + 1: force hi limb below 0xFFFFF00000000000
        a[:, 1] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    return np.ascontiguousarray(a)
