"""MI355X-native T3 speech-token decode engine: drop-in for the vLLM engine behind
ChatterboxTTS.generate() (reference src/chatterbox_vllm/tts.py:150-171, 445-465)."""
from . import constants  # noqa: F401
