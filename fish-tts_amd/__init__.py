"""fish_tts_amd: MI355X-native (gfx950) hot path of smolGura/fish-tts — dual-AR semantic-token decode and
DAC codec decode as hand-written HIP kernels behind the reference's public API
(get_instance / FishTTS.synthesize / synthesize_stream / VoiceProfile)."""
__version__ = "0.1.0"

from .config import CodecArgs, DualARModelArgs, s1_mini_args  # noqa: F401
from .synthesizer import FishTTS, VoiceProfile, get_instance, reset_instance  # noqa: F401
from .tokenizer import ByteTokenizer, TokenLayout  # noqa: F401

__all__ = ["FishTTS", "VoiceProfile", "get_instance", "reset_instance", "DualARModelArgs", "CodecArgs",
           "s1_mini_args", "ByteTokenizer", "TokenLayout"]
