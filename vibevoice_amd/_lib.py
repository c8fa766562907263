"""ctypes binding of libvvhip.so (include/vvhip.h).  There is NO fallback: if the
shared library is missing or a symbol is absent, importing the engine raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VVHIP_LIB", os.path.join(HERE, "libvvhip.so"))   # override only for A/B benchmarking of builds


class VVConfig(C.Structure):
    _fields_ = [
        ("lm_hidden", C.c_int), ("lm_layers", C.c_int), ("lm_heads", C.c_int), ("lm_kv_heads", C.c_int),
        ("lm_head_dim", C.c_int), ("lm_inter", C.c_int), ("lm_vocab", C.c_int), ("lm_eps", C.c_float),
        ("head_layers", C.c_int), ("head_ffn", C.c_int), ("latent_dim", C.c_int), ("head_eps", C.c_float),
        ("n_filters", C.c_int), ("n_ratios", C.c_int), ("ratios", C.c_int * 8),
        ("n_stages", C.c_int), ("enc_depths", C.c_int * 8), ("sem_dim", C.c_int),
        ("has_acoustic_encoder", C.c_int), ("codec_eps", C.c_float),
        ("n_slots", C.c_int), ("max_ctx", C.c_int), ("max_rows", C.c_int), ("xsplit", C.c_int),
        ("attn_splits", C.c_int), ("enc_frames", C.c_int), ("use_graph", C.c_int), ("tts_layers", C.c_int),
    ]


class VVRow(C.Structure):
    _fields_ = [("cache", C.c_int), ("pos", C.c_int)]


_P = C.c_void_p
_SIGS = {
    "vv_create": (C.c_int, [C.POINTER(VVConfig), C.POINTER(_P)]),
    "vv_create_shared": (C.c_int, [C.POINTER(VVConfig), _P, C.POINTER(_P)]),
    "vv_destroy": (None, [_P]),
    "vv_last_error": (C.c_char_p, [_P]),
    "vv_num_weights": (C.c_int, [_P]),
    "vv_weight_info": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "vv_upload": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int64]),
    "vv_set_speech_factors": (C.c_int, [_P, C.c_float, C.c_float]),
    "vv_set_valid_tokens": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int]),
    "vv_set_schedule": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "vv_set_schedule_sde": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "vv_lm_forward": (C.c_int, [_P, _P, C.c_int, C.POINTER(VVRow), _P, _P]),
    "vv_lm_forward_range": (C.c_int, [_P, _P, C.c_int, C.POINTER(VVRow), _P, _P, C.c_int, C.c_int, C.c_int]),
    "vv_kv_import": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]),
    "vv_kv_import_at": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]),
    "vv_add_type_embedding": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P]),
    "vv_eos_logit": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "vv_embed": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int), _P]),
    "vv_lm_logits": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "vv_lm_logits_full": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "vv_diffusion_sample": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_float, _P]),
    "vv_diffusion_sample_sde": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_float, _P]),
    "vv_head_forward": (C.c_int, [_P, _P, C.c_int, _P, C.POINTER(C.c_float), _P, _P]),
    "vv_codec_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_int]),
    "vv_semantic_encode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "vv_codec_chain_batch": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, C.c_int]),
    "vv_acoustic_encode": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "vv_acoustic_encode_ragged": (C.c_int, [_P, _P, C.c_int, C.c_longlong, _P, _P]),
    "vv_kv_move": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "vv_set_enc_pass_frames": (C.c_int, [_P, C.c_int]),
    "vv_audio_to_pcm16": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "vv_codec_reset": (C.c_int, [_P, _P, C.c_int]),
    "vv_connect": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "vv_packed_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "vv_pack_matrix": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "vv_gemm_raw": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              _P, C.c_float, _P, _P, C.c_int, C.c_int, C.c_int]),
    "vv_gemm3_raw": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_float, _P, _P, _P, _P]),
    "vv_profile_begin": (C.c_int, [_P]),
    "vv_profile_end": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vv_profile_replay": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vv_profile_replay_family": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vv_stat": (C.c_int64, [_P, C.c_int]),
    "vv_build_id": (C.c_char_p, []),
    "vv_check": (C.c_int, [_P, _P]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def load():
    """dlopen libvvhip.so and bind every symbol of include/vvhip.h; raises if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built "
            "(run `python -m vibevoice_amd.build`); there is no CPU/PyTorch fallback")
    from . import build as _build
    if _build.have_sources() and _build.binary_id(LIB_PATH) != _build.source_id():
        # a binary from other sources must never run silently: rebuild where a compiler exists, refuse otherwise
        import sys
        print(f"[vibevoice_amd] {LIB_PATH} was built from other sources (binary id {_build.binary_id(LIB_PATH)}, sources "
              f"{_build.source_id()}): rebuilding with hipcc (about two minutes)", file=sys.stderr, flush=True)
        try:
            _build.build_locked()          # one process compiles; the other ranks of a torchrun job wait for its binary
        except Exception as ex:
            raise RuntimeError(
                f"{LIB_PATH} was built from different sources (binary id {_build.binary_id(LIB_PATH)}, sources "
                f"{_build.source_id()}) and could not be rebuilt ({ex}); run `python -m vibevoice_amd.build`") from ex
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
