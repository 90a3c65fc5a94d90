"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/mi_speech.h declares, and fails LOUDLY (status + message, no fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name="mi_speech.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mis_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    l = C.CDLL(_lib.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(l, name), f"{name} declared in mi_speech.h but not exported"
    # the Python binding table covers exactly the header
    assert sorted(_lib.SYMBOLS) == declared
    # test scaffolding lives in its own header and its own table: nothing named mis_debug_* is part of the product surface
    assert not [n for n in declared if n.startswith("mis_debug_")]
    debug = _header_symbols("mi_speech_debug.h")
    assert debug and all(n.startswith("mis_debug_") for n in debug) and sorted(_lib.DEBUG_SYMBOLS) == debug
    for name in debug:
        assert hasattr(l, name), f"{name} declared in mi_speech_debug.h but not exported"


def test_abi_version_and_struct_layouts():
    assert _lib.lib().mis_abi_version() == 1
    assert C.sizeof(_lib.SnacConfigC) == 4 * (4 + 8 + 3 + 8 + 3)
    assert C.sizeof(_lib.GenParamsC) == 56
    assert C.sizeof(_lib.LmConfigC) == 23 * 4


@pytest.mark.skipif(_lib.lib().mis_device_count() > 0, reason="GPU present")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(mas.AudioGenerationError) as e:
        mas.SNAC(mas.SNACConfig())
    assert e.value.case == "device"
    with pytest.raises(mas.AudioGenerationError):
        mas.deinterleave(np.zeros(7, np.int32))
    with pytest.raises(mas.AudioGenerationError):
        mas.LlamaTTSModel(mas.LlamaTTSConfiguration(hidden_size=64, num_hidden_layers=1, intermediate_size=64,
                                                    num_attention_heads=1, num_key_value_heads=1, head_dim=64,
                                                    vocab_size=100))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mlx-audio-swift_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle", txt, flags=re.M), f
            if f.endswith((".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"#include\s+[\"<].*oracle", txt), f


def test_config_mirrors():
    c = mas.LlamaTTSConfiguration.from_dict({"hidden_size": 3072, "num_hidden_layers": 28, "intermediate_size": 8192,
                                             "num_attention_heads": 24, "num_key_value_heads": 8, "rms_norm_eps": 1e-5,
                                             "vocab_size": 156940, "rope_theta": 500000.0,
                                             "rope_scaling": {"factor": 32.0, "rope_type": "llama3"}})
    cc = c.to_c()
    assert (cc.hidden_size, cc.num_key_value_heads, cc.tie_word_embeddings) == (3072, 8, 1)   # tie default true
    assert cc.rope_low_freq_factor == 1.0 and cc.rope_high_freq_factor == 4.0
    with pytest.raises(mas.AudioGenerationError):
        mas.LlamaTTSConfiguration.from_dict({"rope_scaling": {"rope_type": "llama3"}})       # factor missing
    s = mas.SNACConfig().to_c()
    assert s.latent_dim == 768 and list(s.decoder_rates)[:4] == [8, 8, 4, 2] and list(s.vq_strides)[:3] == [4, 2, 1]
    p = mas.GenerateParameters()
    assert (p.max_tokens, p.temperature, p.top_p, p.repetition_penalty, p.repetition_context_size) == (1200, 0.6, 0.8, 1.3, 20)
    assert (mas.OrpheusTokens.start_of_speech, mas.OrpheusTokens.audio_token_offset) == (128257, 128266)


def test_split_k_cost_model_picks_the_measured_optima():
    """gemm_choose_split: busiest CU's share x whole-group load waste x slab overhead.  Orpheus-3B at batch 32
    (R = 2 n-tiles per item, 4 waves per item, 256 CUs): qkv 160 items x 96 k-tiles -> S = 3 (8 k-tiles per wave, 480 blocks),
    o_proj 96 x 96 -> S = 2, down 96 x 256 -> S = 8 - the optima of every sweep in profiles/r01_v{1,2,5}_gemm_sweep.json."""
    L = _lib.lib()
    assert L.mis_debug_choose_split(160, 96, 4, 8) == 3
    assert L.mis_debug_choose_split(96, 96, 4, 16) == 2
    assert L.mis_debug_choose_split(96, 256, 4, 16) == 8
    assert L.mis_debug_choose_split(4, 2, 4, 16) == 1                 # tiny model: K too short to split
    assert L.mis_debug_choose_split(0, 96, 4, 8) == 0                 # invalid arguments
