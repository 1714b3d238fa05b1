"""ctypes binding of libfishtts_hip.so (include/fishtts_hip.h).  No torch types cross this
boundary: pointers and sizes only.  Loading fails loudly when the HIP library is missing —
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libfishtts_hip.so")

FT_OK = 0
FT_ERR_ARG, FT_ERR_HIP, FT_ERR_STATE, FT_ERR_UNSUPPORTED, FT_ERR_NOMEM, FT_ERR_TOO_LONG, FT_ERR_MISSING_WEIGHT = range(1, 8)
FT_F32, FT_BF16, FT_F16 = 0, 1, 2


class ft_ar_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "vocab_size", "n_layer", "n_head", "dim", "intermediate_size",
                                         "n_local_heads", "head_dim")] + \
               [("rope_base", C.c_float), ("norm_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("max_seq_len", "tie_word_embeddings", "attention_qkv_bias", "attention_o_bias",
                                         "attention_qk_norm", "codebook_size", "num_codebooks",
                                         "scale_codebook_embeddings", "n_fast_layer", "fast_dim", "fast_n_head",
                                         "fast_n_local_heads", "fast_head_dim", "fast_intermediate_size",
                                         "fast_attention_qkv_bias", "fast_attention_qk_norm", "fast_attention_o_bias",
                                         "semantic_begin_id", "semantic_end_id", "im_end_id", "max_batch",
                                         "max_new_tokens")]


class ft_codec_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dtype", "n_codebooks", "codebook_size", "semantic_codebook_size",
                                         "codebook_dim", "latent_dim", "n_tf_layer", "tf_n_head", "tf_head_dim",
                                         "tf_ffn", "tf_window")] + \
               [("tf_rope_base", C.c_float), ("tf_norm_eps", C.c_float)] + \
               [("n_upsample", C.c_int32), ("decoder_dim", C.c_int32), ("n_rates", C.c_int32),
                ("rates", C.c_int32 * 8), ("max_frames", C.c_int32), ("max_batch", C.c_int32),
                ("encoder_dim", C.c_int32), ("n_enc_rates", C.c_int32), ("enc_rates", C.c_int32 * 8),
                ("enc_tf_layers", C.c_int32 * 8), ("enc_tf_window", C.c_int32), ("max_enc_frames", C.c_int32)]


class ft_sampling(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("repetition_penalty", C.c_float),
                ("ban_eos", C.c_int32), ("seed", C.c_uint64)]


# every symbol include/fishtts_hip.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "ft_create": (C.c_int32, [C.POINTER(ft_ar_config), C.POINTER(ft_codec_config), C.c_int32, C.POINTER(_P)]),
    "ft_destroy": (None, [_P]),
    "ft_last_error": (C.c_char_p, [_P]),
    "ft_load_weight": (C.c_int32, [_P, C.c_char_p, _P, C.c_int32, C.POINTER(C.c_int64), C.c_int32]),
    "ft_finalize_weights": (C.c_int32, [_P]),
    "ft_ar_reset": (C.c_int32, [_P, C.c_int32]),
    "ft_ar_prefill": (C.c_int32, [_P, C.c_int32, _P, C.c_int32, C.POINTER(ft_sampling), _P]),
    "ft_ar_park": (C.c_int32, [_P, C.c_int32]),
    "ft_ar_prefill_slow": (C.c_int32, [_P, C.c_int32, _P, C.c_int32, C.c_int32]),
    "ft_ar_first_frames": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(ft_sampling), _P, _P]),
    "ft_ar_prefill_slow_many": (C.c_int32, [_P, C.c_int32, _P, _P, _P, _P]),
    "ft_ar_prefill_at": (C.c_int32, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.POINTER(ft_sampling), _P]),
    "ft_ar_kv_save": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "ft_ar_kv_restore": (C.c_int32, [_P, _P, C.c_int32]),
    "ft_ar_kv_positions": (C.c_int32, [_P]),
    "ft_ar_kv_free": (None, [_P, _P]),
    "ft_ar_decode": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(ft_sampling), C.c_int32, _P, _P]),
    "ft_ar_set_noise": (C.c_int32, [_P, _P, C.c_int64, C.c_int64]),
    "ft_ar_get_debug": (C.c_int32, [_P, C.c_int32, _P, _P]),
    "ft_codec_decode": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "ft_codec_frame_len": (C.c_int32, [_P]),
    "ft_codec_stream_begin": (C.c_int32, [_P, C.POINTER(_P)]),
    "ft_codec_stream_decode": (C.c_int32, [_P, _P, _P, C.c_int32, _P]),
    "ft_codec_stream_end": (None, [_P, _P]),
    "ft_codec_encode": (C.c_int32, [_P, _P, C.c_int64, _P, _P]),
    "ft_codec_enc_frame_len": (C.c_int32, [_P]),
    "ft_codec_rvq_encode": (C.c_int32, [_P, _P, C.c_int32, _P]),
    "ft_ar_profile_gemv": (C.c_int32, [_P, C.c_int32, C.POINTER(ft_sampling), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ft_ar_profile_frame": (C.c_int32, [_P, C.c_int32, C.POINTER(ft_sampling), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_int32)]),
    "ft_sync": (C.c_int32, [_P]),
    "ft_ar_engine_state": (C.c_int32, [_P, _P, _P, _P]),
    "ft_ar_frame_path": (C.c_char_p, [_P]),
    "ft_test_engine_fault": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "ft_test_sample": (C.c_int32, [_P, _P, C.c_int32, C.POINTER(ft_sampling), _P, _P, _P]),
}

HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-Wno-unused-value"]
SOURCES = ["engine.hip", "codec.hip"]


def _deps(path: str, seen=None) -> set:
    """The source plus every local header it includes (recursively)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        for line in f:
            if line.startswith("#include \""):
                _deps(os.path.join(os.path.dirname(path), line.split('"')[1]), seen)
    return seen


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree with hipcc for gfx950 (cross-compiles without a GPU): one object per source,
    stale ones rebuilt side by side, then linked."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        newest = max(os.path.getmtime(d) for d in _deps(src))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            jobs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in jobs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out[-12000:])
    if jobs or not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + " ".join(cmd) + "\n" + r.stdout[-4000:] + r.stderr[-8000:])
    return LIB_PATH


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the MI355X path has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
