"""ctypes binding of libmi_speech.so (include/mi_speech.h).  Fails loudly when the library is
missing - there is no fallback path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIS_LIB_PATH: load a diagnostics build of the same library instead (e.g. `make timing`, tools/attn_phases.py)
LIB_PATH = os.environ.get("MIS_LIB_PATH") or os.path.join(_HERE, "libmi_speech.so")

MIS_OK = 0
STATUS_NAMES = {0: "ok", 1: "modelNotInitialized", 2: "generationFailed", 3: "invalidInput",
                4: "audioDecodingFailed", 5: "audioEncodingFailed", 6: "cancelled", 7: "device"}
MIS_F32, MIS_F16, MIS_BF16, MIS_I32 = 0, 1, 2, 3
EVENT_TOKEN, EVENT_INFO, EVENT_AUDIO = 0, 1, 2


class SnacConfigC(C.Structure):
    _fields_ = [("sampling_rate", C.c_int32), ("latent_dim", C.c_int32), ("decoder_dim", C.c_int32),
                ("n_decoder_rates", C.c_int32), ("decoder_rates", C.c_int32 * 8), ("codebook_size", C.c_int32),
                ("codebook_dim", C.c_int32), ("n_codebooks", C.c_int32), ("vq_strides", C.c_int32 * 8),
                ("noise", C.c_int32), ("depthwise", C.c_int32), ("attn_window_size", C.c_int32)]


class LmConfigC(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_hidden_layers", C.c_int32), ("intermediate_size", C.c_int32),
                ("num_attention_heads", C.c_int32), ("num_key_value_heads", C.c_int32), ("head_dim", C.c_int32),
                ("vocab_size", C.c_int32), ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float),
                ("rope_factor", C.c_float), ("rope_low_freq_factor", C.c_float), ("rope_high_freq_factor", C.c_float),
                ("rope_original_max_pos", C.c_float), ("tie_word_embeddings", C.c_int32), ("sample_rate", C.c_int32),
                ("qk_norm", C.c_int32), ("rope_plain", C.c_int32), ("rope_ops_in_dtype", C.c_int32),
                ("start_of_speech_id", C.c_int32), ("end_of_speech_id", C.c_int32), ("audio_token_offset", C.c_int32),
                ("start_of_ai_id", C.c_int32), ("codec_chunk_groups", C.c_int32)]


class GenParamsC(C.Structure):
    _fields_ = [("max_tokens", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float),
                ("repetition_penalty", C.c_float), ("repetition_context", C.c_int32), ("seed", C.c_uint64),
                ("frame_constrained", C.c_int32), ("row_offset", C.c_int64),
                ("sampler_flavor", C.c_int32), ("reserved", C.c_int32)]


class GenInfoC(C.Structure):
    _fields_ = [("prompt_token_count", C.c_int32), ("generation_token_count", C.c_int32),
                ("prefill_time", C.c_double), ("generate_time", C.c_double), ("tokens_per_second", C.c_double),
                ("peak_memory_gb", C.c_double)]


class SopranoConfigC(C.Structure):
    _fields_ = [("lm", LmConfigC), ("decoder_num_layers", C.c_int32), ("decoder_dim", C.c_int32),
                ("decoder_intermediate_dim", C.c_int32), ("hop_length", C.c_int32), ("n_fft", C.c_int32),
                ("upscale", C.c_int32), ("input_kernel", C.c_int32), ("dw_kernel", C.c_int32),
                ("token_size", C.c_int32), ("stop_token_id", C.c_int32)]


class Qwen3TTSConfigC(C.Structure):
    _fields_ = [("talker", LmConfigC), ("predictor", LmConfigC), ("num_code_groups", C.c_int32),
                ("text_hidden_size", C.c_int32), ("text_vocab_size", C.c_int32), ("codec_eos_token_id", C.c_int32),
                ("tts_pad_token_id", C.c_int32),
                ("dec_latent_dim", C.c_int32), ("dec_codebook_dim", C.c_int32), ("dec_codebook_size", C.c_int32),
                ("dec_decoder_dim", C.c_int32), ("dec_hidden_size", C.c_int32), ("dec_intermediate_size", C.c_int32),
                ("dec_head_dim", C.c_int32), ("dec_num_heads", C.c_int32), ("dec_num_layers", C.c_int32),
                ("dec_num_kv_heads", C.c_int32), ("dec_num_quantizers", C.c_int32), ("dec_num_semantic_quantizers", C.c_int32),
                ("dec_rms_norm_eps", C.c_float), ("dec_rope_theta", C.c_float),
                ("n_upsample_rates", C.c_int32), ("upsample_rates", C.c_int32 * 8),
                ("n_upsampling_ratios", C.c_int32), ("upsampling_ratios", C.c_int32 * 8), ("sample_rate", C.c_int32)]


class Qwen3TTSReferenceConfigC(C.Structure):
    _fields_ = [("spk_mel_dim", C.c_int32), ("spk_enc_dim", C.c_int32), ("spk_n_blocks", C.c_int32),
                ("spk_channels", C.c_int32 * 8), ("spk_kernel_sizes", C.c_int32 * 8), ("spk_dilations", C.c_int32 * 8),
                ("spk_attention_channels", C.c_int32), ("spk_res2net_scale", C.c_int32), ("spk_se_channels", C.c_int32),
                ("spk_sample_rate", C.c_int32),
                ("enc_audio_channels", C.c_int32), ("enc_num_filters", C.c_int32), ("enc_kernel_size", C.c_int32),
                ("enc_last_kernel_size", C.c_int32), ("enc_residual_kernel_size", C.c_int32),
                ("enc_num_residual_layers", C.c_int32), ("enc_dilation_growth_rate", C.c_int32), ("enc_compress", C.c_int32),
                ("enc_n_ratios", C.c_int32), ("enc_upsampling_ratios", C.c_int32 * 8),
                ("enc_use_causal_conv", C.c_int32), ("enc_use_conv_shortcut", C.c_int32),
                ("enc_hidden_size", C.c_int32), ("enc_num_layers", C.c_int32), ("enc_num_heads", C.c_int32),
                ("enc_intermediate_size", C.c_int32),
                ("enc_codebook_dim", C.c_int32), ("enc_codebook_size", C.c_int32), ("enc_num_quantizers", C.c_int32),
                ("enc_valid_num_quantizers", C.c_int32), ("enc_sampling_rate", C.c_int32),
                ("enc_rope_theta", C.c_float), ("enc_frame_rate", C.c_float), ("enc_norm_eps", C.c_float)]


class Qwen3TTSParamsC(C.Structure):
    _fields_ = [("max_frames", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
                ("repetition_penalty", C.c_float), ("min_p", C.c_float), ("seed", C.c_uint64), ("row_offset", C.c_int64)]


class DacConfigC(C.Structure):
    _fields_ = [("latent_dim", C.c_int32), ("decoder_dim", C.c_int32), ("n_decoder_rates", C.c_int32),
                ("decoder_rates", C.c_int32 * 8), ("n_codebooks", C.c_int32), ("codebook_size", C.c_int32),
                ("codebook_dim", C.c_int32), ("sample_rate", C.c_int32)]


class EncodecConfigC(C.Structure):
    _fields_ = [("audio_channels", C.c_int32), ("num_filters", C.c_int32), ("kernel_size", C.c_int32),
                ("num_residual_layers", C.c_int32), ("dilation_growth_rate", C.c_int32), ("codebook_size", C.c_int32),
                ("codebook_dim", C.c_int32), ("hidden_size", C.c_int32), ("num_lstm_layers", C.c_int32),
                ("residual_kernel_size", C.c_int32), ("use_causal_conv", C.c_int32), ("pad_reflect", C.c_int32),
                ("last_kernel_size", C.c_int32), ("compress", C.c_int32), ("use_conv_shortcut", C.c_int32),
                ("trim_right_ratio", C.c_float), ("n_upsampling_ratios", C.c_int32), ("upsampling_ratios", C.c_int32 * 8),
                ("n_quantizers", C.c_int32), ("sampling_rate", C.c_int32), ("group_norm", C.c_int32)]


class MelConfigC(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("hop_length", C.c_int32), ("n_mels", C.c_int32),
                ("window", C.c_int32), ("mel_scale", C.c_int32), ("slaney_norm", C.c_int32), ("drop_last_frame", C.c_int32)]


class WhisperConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "num_mel_bins", "d_model", "encoder_layers",
                                         "encoder_attention_heads", "encoder_ffn_dim", "max_source_positions",
                                         "decoder_layers", "decoder_attention_heads", "decoder_ffn_dim",
                                         "max_target_positions")]


class SttParamsC(C.Structure):
    _fields_ = [("max_tokens", C.c_int32), ("temperature", C.c_float), ("seed", C.c_uint64), ("eot_id", C.c_int32),
                ("timestamp_begin", C.c_int32), ("suppress", C.c_void_p), ("n_suppress", C.c_int32),
                ("begin_suppress", C.c_void_p), ("n_begin_suppress", C.c_int32), ("row_offset", C.c_int64)]


class GroupTimingC(C.Structure):
    _fields_ = [("n_shards", C.c_int32), ("generate_ms", C.c_double), ("slowest_shard_ms", C.c_double), ("gather_ms", C.c_double)]


class TimingC(C.Structure):
    _fields_ = [("prefill_ms", C.c_double), ("decode_ms", C.c_double), ("codec_ms", C.c_double),
                ("step_ms_avg", C.c_double), ("steps", C.c_int32), ("gemm_probe_ms", C.c_double),
                ("gemm_probe_bytes", C.c_double), ("hbm_bytes_per_step", C.c_double)]


EVENT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64)

_P = C.c_void_p
# name -> (restype, argtypes): every symbol include/mi_speech.h declares
SYMBOLS = {
    "mis_last_error": (C.c_char_p, []),
    "mis_abi_version": (C.c_int, []),
    "mis_free": (None, [_P]),
    "mis_device_count": (C.c_int, []),
    "mis_orpheus_deinterleave": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, _P, _P]),
    "mis_speech_parse_output": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mis_orpheus_parse_output": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, _P, _P]),
    "mis_snac_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "mis_snac_create": (C.c_int, [C.POINTER(SnacConfigC), C.c_int, C.POINTER(_P)]),
    "mis_snac_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "mis_snac_finalize": (C.c_int, [_P]),
    "mis_snac_destroy": (None, [_P]),
    "mis_snac_num_samples": (C.c_int64, [_P, C.c_int]),
    "mis_snac_noise_len": (C.c_int64, [_P, C.c_int, C.c_int]),
    "mis_snac_set_noise": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "mis_snac_decode": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.POINTER(_P), _P]),
    "mis_snac_padded_length": (C.c_int64, [_P, C.c_int64]),
    "mis_snac_encode": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P]),
    "mis_snac_debug_tap": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "mis_tts_load": (C.c_int, [C.c_char_p, _P, C.c_int, C.POINTER(_P)]),
    "mis_tts_create": (C.c_int, [C.POINTER(LmConfigC), _P, C.c_int, C.POINTER(_P)]),
    "mis_tts_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "mis_lm_prefill": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "mis_tts_native_quant_bits": (C.c_int, [_P, C.c_int]),
    "mis_tts_init_synthetic_quantized": (C.c_int, [_P, C.c_uint64, C.c_int]),
    "mis_tts_set_tensor_quantized": (C.c_int, [_P, C.c_char_p, _P, _P, _P, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "mis_tts_init_synthetic": (C.c_int, [_P, C.c_uint64]),
    "mis_tts_finalize": (C.c_int, [_P]),
    "mis_tts_destroy": (None, [_P]),
    "mis_lm_reset": (C.c_int, [_P, C.c_int, C.c_int]),
    "mis_lm_forward": (C.c_int, [_P, _P, _P, _P]),
    "mis_lm_forward_hidden": (C.c_int, [_P, _P, _P, _P, _P]),
    "mis_sample_logits": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, C.POINTER(GenParamsC), C.c_int,
                                    C.c_int, C.c_int, _P]),
    "mis_tts_generate": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(GenParamsC), C.POINTER(_P), C.POINTER(_P),
                                   C.POINTER(C.c_int64), _P, C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_tts_generate_device": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(GenParamsC), C.POINTER(_P), _P, C.c_int64,
                                          _P, _P]),
    "mis_tts_generate_stream": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(GenParamsC), C.POINTER(_P), EVENT_CB, _P, _P]),
    "mis_tts_set_profiling": (C.c_int, [_P, C.c_int]),
    "mis_tts_last_timing": (C.c_int, [_P, C.POINTER(TimingC)]),
    "mis_tts_time_gemm": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mis_shard_rows": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mis_tts_group_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(_P)]),
    "mis_tts_group_destroy": (None, [_P]),
    "mis_tts_group_size": (C.c_int, [_P]),
    "mis_tts_group_generate": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(GenParamsC), C.POINTER(_P), C.POINTER(C.c_int64), _P,
                                         C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_tts_group_generate_device": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(GenParamsC), C.POINTER(_P), C.c_int64, _P, _P]),
    "mis_tts_group_last_timing": (C.c_int, [_P, C.POINTER(GroupTimingC)]),
    "mis_comm_unique_id": (C.c_int, [_P]),
    "mis_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "mis_comm_destroy": (None, [_P]),
    "mis_comm_all_gather_pcm": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, C.POINTER(C.c_double)]),
    "mis_mel_num_frames": (C.c_int64, [C.POINTER(MelConfigC), C.c_int64]),
    "mis_mel_spectrogram": (C.c_int, [C.c_int, C.POINTER(MelConfigC), _P, C.c_int, C.c_int64, _P, C.POINTER(C.c_int64)]),
    "mis_whisper_encoder_features": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int64, C.c_int, _P]),
    "mis_whisper_create": (C.c_int, [C.POINTER(WhisperConfigC), C.c_int, C.POINTER(_P)]),
    "mis_whisper_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "mis_whisper_init_synthetic": (C.c_int, [_P, C.c_uint64]),
    "mis_whisper_finalize": (C.c_int, [_P]),
    "mis_whisper_destroy": (None, [_P]),
    "mis_whisper_encode": (C.c_int, [_P, _P, C.c_int, _P]),
    "mis_whisper_decoder_reset": (C.c_int, [_P]),
    "mis_whisper_decoder_forward": (C.c_int, [_P, _P, _P, _P]),
    "mis_stt_whisper_generate": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, C.c_int, C.POINTER(SttParamsC),
                                           C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_stt_whisper_generate_stream": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, C.c_int, C.POINTER(SttParamsC), EVENT_CB, _P, _P,
                                                  C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_soprano_create": (C.c_int, [C.POINTER(SopranoConfigC), C.c_int, C.POINTER(_P)]),
    "mis_soprano_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "mis_soprano_finalize": (C.c_int, [_P]),
    "mis_soprano_destroy": (None, [_P]),
    "mis_soprano_lm": (_P, [_P]),
    "mis_soprano_lm_path": (C.c_int32, [_P]),
    "mis_soprano_num_samples": (C.c_int64, [_P, C.c_int]),
    "mis_soprano_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "mis_soprano_generate": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(GenParamsC), C.POINTER(_P), C.POINTER(C.c_int64),
                                       _P, C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_soprano_generate_stream": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(GenParamsC), EVENT_CB, _P, _P]),
    "mis_qwen3tts_create": (C.c_int, [C.POINTER(Qwen3TTSConfigC), C.c_int, C.POINTER(_P)]),
    "mis_qwen3tts_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "mis_qwen3tts_set_tensor_quantized": (C.c_int, [_P, C.c_char_p, _P, _P, _P, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "mis_qwen3tts_finalize": (C.c_int, [_P]),
    "mis_qwen3tts_destroy": (None, [_P]),
    "mis_qwen3tts_talker": (_P, [_P]),
    "mis_qwen3tts_samples_per_frame": (C.c_int, [_P]),
    "mis_qwen3tts_num_code_groups": (C.c_int, [_P]),
    "mis_qwen3tts_group_generate": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.POINTER(Qwen3TTSParamsC), _P,
                                              C.POINTER(_P), C.POINTER(C.c_int64), _P, C.POINTER(_P), C.POINTER(C.c_int64), _P,
                                              C.c_int, EVENT_CB, _P, _P]),
    "mis_whisper_group_generate": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int64, _P, C.c_int, C.POINTER(SttParamsC),
                                             C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_soprano_group_generate": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.POINTER(GenParamsC), C.POINTER(_P), C.POINTER(C.c_int64),
                                             _P, C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_qwen3tts_generate_codes": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.POINTER(Qwen3TTSParamsC), _P,
                                              C.POINTER(_P), C.POINTER(C.c_int64), _P]),
    "mis_qwen3tts_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "mis_qwen3tts_set_stream_exact": (C.c_int, [_P, C.c_int]),
    "mis_qwen3tts_decode_stream_begin": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "mis_qwen3tts_decode_stream_step": (C.c_int, [_P, _P, C.c_int, _P]),
    "mis_qwen3tts_decode_stream_end": (C.c_int, [_P]),
    "mis_qwen3tts_decoder_tap": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int32),
                                           C.POINTER(C.c_int64)]),
    "mis_qwen3tts_generate": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.POINTER(Qwen3TTSParamsC), _P,
                                        C.POINTER(_P), C.POINTER(C.c_int64), _P, C.POINTER(_P), C.POINTER(C.c_int64), _P,
                                        C.c_int, EVENT_CB, _P, _P]),
    "mis_qwen3tts_enable_reference": (C.c_int, [_P, C.POINTER(Qwen3TTSReferenceConfigC)]),
    "mis_qwen3tts_speaker_embedding": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mis_qwen3tts_encode_audio": (C.c_int, [_P, _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mis_qwen3tts_reference_tap": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, _P, C.c_int64, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int64)]),
    "mis_qwen3tts_add_reference": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mis_qwen3tts_clear_references": (C.c_int, [_P]),
    "mis_qwen3tts_sample_logits": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, C.POINTER(Qwen3TTSParamsC), C.c_int, C.c_int,
                                             C.c_int, C.c_int, _P]),
    "mis_dac_create": (C.c_int, [C.POINTER(DacConfigC), C.c_int, C.POINTER(_P)]),
    "mis_dac_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "mis_dac_finalize": (C.c_int, [_P]),
    "mis_dac_destroy": (None, [_P]),
    "mis_dac_num_samples": (C.c_int64, [_P, C.c_int]),
    "mis_dac_padded_length": (C.c_int64, [_P, C.c_int64]),
    "mis_dac_encode": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, _P, _P]),
    "mis_dac_decode_codes": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "mis_dac_debug_tap": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "mis_mel_stream_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "mis_mel_stream_process": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "mis_mel_stream_flush": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "mis_mel_stream_reset": (C.c_int, [_P]),
    "mis_mel_stream_total_frames": (C.c_int64, [_P]),
    "mis_mel_stream_destroy": (None, [_P]),
    "mis_encodec_create": (C.c_int, [C.POINTER(EncodecConfigC), C.c_int, C.POINTER(_P)]),
    "mis_encodec_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "mis_encodec_finalize": (C.c_int, [_P]),
    "mis_encodec_destroy": (None, [_P]),
    "mis_encodec_hop_length": (C.c_int, [_P]),
    "mis_encodec_decode_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "mis_encodec_debug_tap": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int64)]),
}

# diagnostics / test scaffolding: include/mi_speech_debug.h (not part of the product surface)
DEBUG_SYMBOLS = {
    "mis_debug_launch_floor": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "mis_debug_occupy_cus": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double]),
    "mis_debug_occupy_wait": (C.c_int, []),
    "mis_debug_device_cus": (C.c_int32, [C.c_int]),
    "mis_debug_sampler_failures": (C.c_int32, []),
    "mis_debug_choose_split": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mis_debug_token_engine": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, C.POINTER(C.c_double)]),
}

_lib = None


class MisLibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded shared library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MisLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:        # if torch is installed, let ITS bundled ROCm runtime load first (one libamdhip64 per process)
            import torch  # noqa: F401
        except Exception:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(DEBUG_SYMBOLS.items()):
            fn = getattr(l, name)          # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    m = lib().mis_last_error()
    return m.decode("utf-8", "replace") if m else ""
