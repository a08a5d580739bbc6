"""Token post-filter of the T3 path: drop-in for ``ChatterboxTTS.analyze_and_clean_tokens``
(reference src/chatterbox_vllm/tts.py:300-365) and the range filter of tts.py:514, as one host-side call into the
C ABI (``t3_clean_tokens``) instead of one CUDA tensor + ``.item()`` sync per token."""
from __future__ import annotations

import ctypes as ct
from typing import List, Sequence, Tuple

import numpy as np

from .engine import load_library

REASONS = {0: None, 1: "repetition", 2: "long_tail"}


def analyze_and_clean_tokens(token_ids: Sequence[int], text_token_count: int, range_filter: bool = False) -> Tuple[List[int], str]:
    """token_ids: speech-space ids (offset already removed, tts.py:492); text_token_count: ``len(prompt.split()) * 2``
    as tts.py:496 computes it.  Returns (cleaned ids, reason or None)."""
    ids = np.ascontiguousarray(np.asarray(token_ids, dtype=np.int32))
    out = np.empty(max(1, len(ids)), dtype=np.int32)
    why = ct.c_int32(0)
    n = load_library().t3_clean_tokens(ids.ctypes.data, len(ids), int(text_token_count), 1 if range_filter else 0, out.ctypes.data, ct.byref(why))
    if n < 0:
        raise ValueError(f"t3_clean_tokens failed: {n}")
    return out[:n].tolist(), REASONS[int(why.value)]
