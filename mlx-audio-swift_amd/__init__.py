"""mlx-audio-swift_amd: MI355X-native (gfx950) engine for the TTS generate()/generateStream() + neural
codec hot path of Blaizzy/mlx-audio-swift.

The product is `libmi_speech.so` (hand-written HIP kernels behind the C ABI of include/mi_speech.h,
sources in csrc/).  This package is the host-side mirror of the reference's protocol surface for that
path, written in Python because no Swift toolchain exists in the build image (INTEGRATION.md holds
the Swift shim a maintainer would add):

    codecs.SNAC            <-> class SNAC : AudioCodecModel       (MLXAudioCodecs/SNAC/SNACDecoder.swift)
    codecs.DescriptDAC     <-> class DescriptDAC (decode side)         (MLXAudioCodecs/Descript/DescriptDAC.swift)
    codecs.Encodec         <-> class Encodec (decode side)             (MLXAudioCodecs/Encodec/Encodec.swift)
    tts.LlamaTTSModel      <-> class LlamaTTSModel : SpeechGenerationModel  (MLXAudioTTS/Models/Llama/LlamaTTS.swift)
    soprano.SopranoModel   <-> class SopranoModel : SpeechGenerationModel  (MLXAudioTTS/Models/Soprano/Soprano.swift)
    qwen3tts.Qwen3TTSModel <-> class Qwen3TTSModel : SpeechGenerationModel  (MLXAudioTTS/Models/Qwen3TTS/Qwen3TTS.swift)
    stt.WhisperModel       <-> class WhisperModel : STTGenerationModel   (MLXAudioSTT/Models/Whisper/WhisperModel.swift)
    dsp.*                  <-> computeMelSpectrogram (MLXAudioCore/DSP.swift) / WhisperAudio.encoderFeatures
    generation.*           <-> AudioGeneration / AudioGenerationInfo / AudioGenerationError /
                               GenerateParameters                (MLXAudioCore/Generation/GenerationTypes.swift)

There is NO CPU fallback: importing works anywhere (so that the C ABI can be inspected), but every
compute entry point raises if the HIP library or a GPU is missing.
"""
from . import _lib  # noqa: F401
from .generation import (AudioGenerationError, AudioGenerationInfo, GenerateParameters, TokenEvent, InfoEvent,  # noqa: F401
                         AudioEvent)
from .codecs import SNAC, SNACConfig, DescriptDAC, DescriptDACConfig, Encodec, EncodecConfig  # noqa: F401
from .tts import (LlamaTTSModel, LlamaTTSConfiguration, OrpheusTokens, VyvoTokens, interleave_snac_codes,  # noqa: F401
                  orpheus_prompt_rows, padded_prompt_batch)
from .orpheus import deinterleave, parse_output  # noqa: F401
from .soprano import SopranoModel, SopranoConfiguration  # noqa: F401
from .qwen3tts import (Qwen3TTSModel, Qwen3TTSConfiguration, Qwen3TTSDecoderConfiguration, Qwen3TTSGenerateParameters,  # noqa: F401
                       PreparedPrompt)
from . import dsp  # noqa: F401
from .stt import WhisperModel, WhisperConfig, STTGenerateParameters, STTOutput  # noqa: F401

__all__ = ["SNAC", "SNACConfig", "LlamaTTSModel", "LlamaTTSConfiguration", "OrpheusTokens", "GenerateParameters",
           "AudioGenerationError", "AudioGenerationInfo", "TokenEvent", "InfoEvent", "AudioEvent", "deinterleave",
           "parse_output"]
