"""ctypes binding of libdvae_hip.so (C ABI in include/dvae_hip.h).

The library is built in-tree by ``make -C d-vae_amd/csrc`` (see
``__graft_entry__.build``).  There is no fallback: if the shared object is
missing every entry point raises, it never routes to a CPU path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVAE_LIB: another build of the same library (same-box A-B timing of kernel changes)
LIB_PATH = os.environ.get("DVAE_LIB") or os.path.join(_HERE, "libdvae_hip.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float
c_ll = ctypes.c_longlong


class DvaeDims(ctypes.Structure):
    """Mirror of ``struct dvae_dims``."""
    _fields_ = [(n, c_int) for n in (
        "B", "N", "C", "H", "Z", "mod", "bidir", "npos",
        "HP", "ZP", "EH", "EHP", "VHP", "CP")]


class SvaeDims(ctypes.Structure):
    """Mirror of ``struct dvae_svae_dims``."""
    _fields_ = [(n, c_int) for n in ("B", "T", "C", "N", "H", "Z", "bidir", "I", "IP", "HP", "ZP")]


class PackEntry(ctypes.Structure):
    """Mirror of ``struct dvae_pack_entry``."""
    _fields_ = [("raw", ctypes.c_uint64), ("grad", ctypes.c_int64),
                ("aux_raw", ctypes.c_uint64), ("aux_grad", ctypes.c_int64),
                ("dst_off", ctypes.c_int64),
                ("rows", c_int), ("cols", c_int), ("raw_ld", c_int), ("raw_col0", c_int),
                ("dst_ld", c_int), ("kind", c_int),
                ("g_off", ctypes.c_int64), ("g_ld", c_int), ("g_kind", c_int)]


class DvaeStep(ctypes.Structure):
    """Mirror of ``struct dvae_step``."""
    _fields_ = [("d_enc", ctypes.POINTER(DvaeDims)), ("d_dec", ctypes.POINTER(DvaeDims)),
                ("g", c_void_p), ("x", c_void_p), ("sizes", c_void_p),
                ("sched", c_void_p), ("info_host", c_void_p),
                ("enc_table", c_void_p), ("enc_n", c_int),
                ("dec_table", c_void_p), ("dec_n", c_int),
                ("Pe", c_void_p), ("Pd", c_void_p),
                ("ws_enc", c_void_p), ("ws_dec", c_void_p), ("ws_step", c_void_p),
                ("eps", c_void_p), ("beta", c_float),
                ("Ge", c_void_p), ("Gd", c_void_p), ("out4", c_void_p)]


class DvaeError(RuntimeError):
    pass


class GraphError(AssertionError):
    """Adjacency is not a 0/1 strictly lower-triangular DAG.  The reference
    reports shape/contract violations with ``assert`` (dvae_model.py:62,84,192)."""


# name -> (restype, argtypes); every symbol declared in include/dvae_hip.h
_P = ctypes.POINTER
SIGNATURES = {
    "dvae_last_error": (ctypes.c_char_p, []),
    "dvae_abi_version": (c_int, []),
    "dvae_dims_init": (c_int, [_P(DvaeDims)]),
    "dvae_gemm_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                              c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dvae_gemm_f32_mapped": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                     c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dvae_small_gemm": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_void_p]),
    "dvae_gemm_force_tile": (c_int, [c_int]),
    "dvae_gemm_tune": (c_int, [c_int, c_int]),
    "dvae_gemm_variant": (c_int, [c_int]),
    "dvae_gemm_pf": (c_int, [c_int]),
    "dvae_gemm_x6": (c_int, [c_int]),
    "dvae_gemm_core": (c_int, []),
    "dvae_gemm_force_splits": (c_int, [c_int]),
    "dvae_gemm_dbg": (c_int, [c_int]),
    "dvae_level_tune": (c_int, [c_int, c_int, c_int]),
    "dvae_small_rows_tune": (c_int, [c_int]),
    "dvae_enc_merge_tune": (c_int, [c_int]),
    "dvae_recompute_tune": (c_int, [c_int]),
    "dvae_phase_read": (c_int, [c_void_p, c_void_p, c_int]),
    "dvae_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, ctypes.c_double, ctypes.c_double,
                       ctypes.c_double, ctypes.c_double, ctypes.c_double, c_ll, c_void_p]),
    "dvae_pair_path": (c_int, [c_int, c_int]),
    "dvae_typed_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t,
                          c_void_p]),
    "dvae_wgrad_split_tune": (c_int, [ctypes.c_double, c_int]),
    "dvae_gemm_profile": (c_int, [c_int]),
    "dvae_gemm_profile_read": (c_int, [c_void_p]),
    "dvae_hbm_profile_read": (c_int, [c_void_p]),
    "dvae_sched_bytes": (c_size_t, [c_int, c_int]),
    "dvae_sched_build": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dvae_sched_types": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dvae_sched_export": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dvae_enc_param_floats": (c_size_t, [_P(DvaeDims)]),
    "dvae_dec_param_floats": (c_size_t, [_P(DvaeDims)]),
    "dvae_enc_grad_floats": (c_size_t, [_P(DvaeDims)]),
    "dvae_dec_grad_floats": (c_size_t, [_P(DvaeDims)]),
    "dvae_enc_param_count": (c_int, [_P(DvaeDims)]),
    "dvae_dec_param_count": (c_int, [_P(DvaeDims)]),
    "dvae_enc_pack_table": (c_int, [_P(DvaeDims), c_void_p, c_void_p, c_void_p, c_int]),
    "dvae_dec_pack_table": (c_int, [_P(DvaeDims), c_void_p, c_void_p, c_void_p, c_int]),
    "dvae_pack_params": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "dvae_unpack_grads": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dvae_enc_ws_bytes": (c_size_t, [_P(DvaeDims), c_int, c_int]),
    "dvae_encoder_fwd": (c_int, [_P(DvaeDims)] + [c_void_p] * 12),
    "dvae_encoder_bwd": (c_int, [_P(DvaeDims)] + [c_void_p] * 12),
    "dvae_dec_ws_bytes": (c_size_t, [_P(DvaeDims), c_int, c_int]),
    "dvae_decoder_fwd": (c_int, [_P(DvaeDims)] + [c_void_p] * 12),
    "dvae_decoder_bwd": (c_int, [_P(DvaeDims)] + [c_void_p] * 13),
    "dvae_reparam_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p]),
    "dvae_reparam_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p]),
    "dvae_elbo_ws_bytes": (c_size_t, [c_int]),
    "dvae_elbo": (c_int, [_P(DvaeDims)] + [c_void_p] * 7 + [c_float] + [c_void_p] * 7),
    "dvae_elbo_static": (c_int, [_P(DvaeDims)] + [c_void_p] * 7 + [c_float] + [c_void_p] * 6),
    "dvae_step_ws_bytes": (c_size_t, [_P(DvaeDims)]),
    "dvae_step_part1": (c_int, [_P(DvaeStep), c_void_p, c_void_p]),
    "dvae_step_part2": (c_int, [_P(DvaeStep), c_void_p, c_void_p]),
    "dvae_predict_ws_bytes": (c_size_t, [_P(DvaeDims)]),
    "dvae_decoder_predict": (c_int, [_P(DvaeDims)] + [c_void_p] * 9),
    "dvae_svae_dims_init": (c_int, [_P(SvaeDims)]),
    "dvae_svae_enc_param_floats": (c_size_t, [_P(SvaeDims)]),
    "dvae_svae_dec_param_floats": (c_size_t, [_P(SvaeDims)]),
    "dvae_svae_enc_param_count": (c_int, [_P(SvaeDims)]),
    "dvae_svae_dec_param_count": (c_int, [_P(SvaeDims)]),
    "dvae_svae_enc_pack_table": (c_int, [_P(SvaeDims), c_void_p, c_void_p, c_void_p, c_int]),
    "dvae_svae_dec_pack_table": (c_int, [_P(SvaeDims), c_void_p, c_void_p, c_void_p, c_int]),
    "dvae_svae_enc_ws_bytes": (c_size_t, [_P(SvaeDims)]),
    "dvae_svae_dec_ws_bytes": (c_size_t, [_P(SvaeDims)]),
    "dvae_svae_enc_fwd": (c_int, [_P(SvaeDims)] + [c_void_p] * 6),
    "dvae_svae_enc_bwd": (c_int, [_P(SvaeDims)] + [c_void_p] * 7),
    "dvae_svae_dec_fwd": (c_int, [_P(SvaeDims)] + [c_void_p] * 6),
    "dvae_svae_dec_bwd": (c_int, [_P(SvaeDims)] + [c_void_p] * 8),
    "dvae_svae_elbo": (c_int, [_P(SvaeDims)] + [c_void_p] * 5 + [c_float] + [c_void_p] * 7),
    "dvae_svae_predict": (c_int, [_P(SvaeDims), c_ll] + [c_void_p] * 7),
    "dvae_side_stream_create": (c_int, [_P(c_void_p), _P(c_int)]),
    "dvae_side_stream_destroy": (c_int, [c_void_p]),
}

_lib = None


def load():
    """Load the shared library once; raises DvaeError if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DvaeError(
            "libdvae_hip.so not found at %s: build it with `make -C d-vae_amd/csrc` "
            "(there is no CPU fallback for the D-VAE hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status == 0:
        return
    msg = load().dvae_last_error().decode("utf-8", "replace")
    if status == -3:
        raise GraphError(msg)
    raise DvaeError("libdvae_hip status %d: %s" % (status, msg))


def make_dims(B, N, C, H, Z, mod, bidir, npos):
    d = DvaeDims()
    d.B, d.N, d.C, d.H, d.Z = int(B), int(N), int(C), int(H), int(Z)
    d.mod = {None: 0, "bn": 1}[mod]
    d.bidir = int(bool(bidir))
    d.npos = int(npos) if mod is None else 0
    check(load().dvae_dims_init(ctypes.byref(d)))
    return d


def make_svae_dims(B, T, C, N, H, Z, bidir):
    d = SvaeDims()
    d.B, d.T, d.C, d.N, d.H, d.Z, d.bidir = int(B), int(T), int(C), int(N), int(H), int(Z), int(bool(bidir))
    check(load().dvae_svae_dims_init(ctypes.byref(d)))
    return d


def sched_info_ints(N):
    return 8 + 6 * (N + 1) + 4


def sched_merge_at(N):
    """First int of the shared-encoder-row summary (include/dvae_hip.h)."""
    return 8 + 4 * (N + 1)
