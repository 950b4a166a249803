"""ctypes binding of libdetail_hip.so (include/detail_hip.h).  Fails loudly when the library is missing —
there is NO CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DTTS_LIB_PATH: another build of the same library (diagnostic builds of tools/diag_token_pk.py); the default is the in-tree product library
LIB_PATH = os.environ.get("DTTS_LIB_PATH") or os.path.join(HERE, "libdetail_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_u64_p = C.POINTER(C.c_ulonglong)


class DttsConfig(C.Structure):
    _fields_ = [
        ("diff_channels", C.c_int), ("diff_layers", C.c_int), ("diff_heads", C.c_int), ("mel_channels", C.c_int),
        ("diff_out_channels", C.c_int), ("diff_steps", C.c_int), ("diff_trained_steps", C.c_int), ("cond_free_k", C.c_float),
        ("gpt_dim", C.c_int), ("gpt_layers", C.c_int), ("gpt_heads", C.c_int), ("gpt_mel_codes", C.c_int),
        ("gpt_text_tokens", C.c_int), ("gpt_max_mel_pos", C.c_int), ("gpt_max_text_pos", C.c_int),
        ("inter_channels", C.c_int), ("hidden_channels", C.c_int), ("filter_channels", C.c_int), ("enc_heads", C.c_int),
        ("enc_layers", C.c_int), ("gin_channels", C.c_int), ("upsample_initial_channel", C.c_int), ("n_upsamples", C.c_int),
        ("upsample_rates", C.c_int * 8), ("upsample_kernels", C.c_int * 8), ("n_resblock_kernels", C.c_int),
        ("resblock_kernels", C.c_int * 4), ("resblock_dilations", C.c_int * 4),
    ]


class DttsGptOptions(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("seed", C.c_ulonglong), ("sample_ids", c_int_p), ("max_generate_length", C.c_int), ("top_k", C.c_int),
                ("top_p", C.c_float), ("temperature", C.c_float), ("repetition_penalty", C.c_float), ("suppress_eos", C.c_int),
                ("forced_uniforms", C.c_void_p), ("forced_codes", c_int_p), ("row_seeds", c_u64_p), ("typical_mass", C.c_float), ("token_wgs", C.c_int)]


class DttsKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_longlong), ("total_ms", C.c_double), ("union_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# name -> (restype, argtypes); every symbol include/detail_hip.h declares
SIGNATURES = {
    "dtts_version": (C.c_char_p, []),
    "dtts_default_config": (None, [C.POINTER(DttsConfig)]),
    "dtts_gpt_options_init": (None, [C.POINTER(DttsGptOptions)]),
    "dtts_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(DttsConfig), C.c_int]),
    "dtts_destroy": (C.c_int, [C.c_void_p]),
    "dtts_last_error": (C.c_char_p, [C.c_void_p]),
    "dtts_bind_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p), c_u64_p, c_u64_p, C.c_int, C.c_void_p]),
    "dtts_vq_decode": (C.c_int, [C.c_void_p, c_int_p, c_int_p, C.c_int, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_vq_encode": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dtts_resample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_mel_spectrogram": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_spectrogram": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "dtts_profile_enable": (C.c_int, [C.c_int]),
    "dtts_profile_sampling": (C.c_int, [C.c_int]),
    "dtts_profile_report": (C.c_int, [C.POINTER(DttsKernelStat), C.c_int]),
    "dtts_gpt_generate": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int, C.c_int,
                                    C.POINTER(DttsGptOptions), c_int_p, c_int_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_gpt_prefill": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int, C.c_int,
                                   C.POINTER(DttsGptOptions), C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_gpt_decode_step": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dtts_gpt_decode": (C.c_int, [C.c_void_p, C.c_int, c_int_p, C.c_void_p]),
    "dtts_gpt_steps": (C.c_int, [C.c_void_p]),
    "dtts_gpt_all_finished": (C.c_int, [C.c_void_p, c_int_p, C.c_void_p]),
    "dtts_gpt_finish": (C.c_int, [C.c_void_p, c_int_p, c_int_p, C.c_void_p]),
    "dtts_op_sample_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_int_p, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                        C.c_float, C.c_float, c_int_p, C.c_void_p]),
    "dtts_diff_p_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_ulonglong, c_int_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "dtts_gpt_latents": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int,
                                   C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_diff_conditioning": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_diff_timestep_independent": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dtts_diff_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_diff_sample": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_ulonglong, c_int_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_vocoder": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_ulonglong, c_int_p, C.c_float, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "dtts_vocoder_stream": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_ulonglong, c_int_p, C.c_float, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_vocoder_ticket": (C.c_longlong, [C.c_void_p]),
    "dtts_vocoder_check": (C.c_int, [C.c_void_p, C.c_longlong]),
    "dtts_vocoder_check_active": (C.c_int, [C.c_void_p]),
    "dtts_generator": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_op_mel_style": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_op_attention_block": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_op_resblock": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_op_resblock1": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_op_wn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dtts_op_enc_p": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dtts_op_conv1d": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_int_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dtts_op_philox_normal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_ulonglong, c_int_p, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


class LibraryMissing(RuntimeError):
    pass


def load():
    """dlopen libdetail_hip.so.  `import torch` first so our NEEDED libamdhip64.so binds to the HIP runtime torch
    already loaded (SURVEY.md §7 toolchain facts) and torch's streams are valid inside the library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found — build it with `python -m detail_tts_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the product path.")
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
