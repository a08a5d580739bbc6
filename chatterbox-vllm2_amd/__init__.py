"""MI355X-native T3 speech-token decode engine: drop-in for the vLLM engine behind
ChatterboxTTS.generate() (reference src/chatterbox_vllm/tts.py:150-171, 445-465).

    from chatterbox_vllm2_amd import LLM, SamplingParams        # instead of `from vllm import ...`
"""
from . import constants  # noqa: F401
from .constants import SPEECH_TOKEN_OFFSET  # noqa: F401


def __getattr__(name):          # lazy: importing the package must not need the built library
    if name in ("LLM", "SamplingParams", "RequestOutput", "CompletionOutput"):
        from . import llm
        return getattr(llm, name)
    if name in ("T3Engine",):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
