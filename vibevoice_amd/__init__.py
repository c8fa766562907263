"""vibevoice_amd -- MI355X-native engine for the VibeVoice generate() hot path.

    from vibevoice_amd import VibeVoiceForConditionalGenerationInference
    from vibevoice_amd import VibeVoiceStreamingForConditionalGenerationInference
"""
from .engine import Engine, EngineConfig, EngineError  # noqa: F401
from .modeling import VibeVoiceForConditionalGenerationInference, VibeVoiceGenerationOutput  # noqa: F401
from .modeling_streaming import VibeVoiceStreamingForConditionalGenerationInference  # noqa: F401
from .lora import load_lora_assets  # noqa: F401
from .streamer import AsyncAudioStreamer, AudioStreamer  # noqa: F401
