"""Prompt front-end of the T3 path: text -> token ids -> the <cond | text | BOS> id layout.

Reference: T3MultiModalProcessor.apply (src/chatterbox_vllm/models/t3/t3.py:143-249) builds
``[695, ids[0] x 32, 696, *ids, 697]`` (t3.py:189-200) and a [T,1024] multimodal tensor
``cond_emb(34) | lower-triangular ones | zeros(1)`` (t3.py:212-221) whose only job is to carry text
positions across vLLM's chunked prefill.  This engine's scheduler knows every row's position, so only
the id layout is needed at the C ABI; ``build_mm_tensor`` is kept for data-contract fixtures.

Tokenizers: EnTokenizer._tokenize (models/t3/entokenizer.py:67-69) and MTLTokenizer._tokenize
(models/t3/mtltokenizer.py:300-327) over the reference's own ``tokenizers`` JSON files, which are data
files of the reference checkout and are NOT copied into this repository: pass their location.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence
from unicodedata import normalize

import torch

from . import constants as C

SPACE = "[SPACE]"


def assemble_prompt_ids(text_ids: Sequence[int]) -> List[int]:
    """t3.py:189-200."""
    if len(text_ids) == 0:
        raise ValueError("empty text prompt")
    return ([C.PREFILL_COND_START_TOKEN] + [int(text_ids[0])] * (C.CONDITIONING_SIZE - 2) + [C.PREFILL_COND_END_TOKEN]
            + [int(t) for t in text_ids] + [C.PREFILL_END_TOKEN])


def build_mm_tensor(cond_emb: torch.Tensor, n_text: int) -> torch.Tensor:
    """t3.py:212-221: [34 + n_text + 1, 1024]."""
    tri = (torch.arange(cond_emb.shape[1]).unsqueeze(0) <= torch.arange(n_text).unsqueeze(1)).float()
    return torch.cat([cond_emb, tri.to(cond_emb.device), torch.zeros(1, cond_emb.shape[1], device=cond_emb.device)], dim=0)


_SEARCH_DIRS = ["", "src/chatterbox_vllm/models/t3"]
_FILES = {"EnTokenizer": "tokenizer.json", "MtlTokenizer": "grapheme_mtl_merged_expanded_v1.json"}


def find_tokenizer_file(kind: str, model_dir: Optional[str] = None, explicit: Optional[str] = None) -> Optional[str]:
    if explicit:
        return explicit
    fname = _FILES[kind]
    roots = [os.environ.get("CHATTERBOX_TOKENIZER_DIR"), model_dir, os.getcwd()]
    for r in roots:
        if not r:
            continue
        for d in _SEARCH_DIRS:
            p = os.path.join(r, d, fname)
            if os.path.exists(p):
                return p
    return None


# text clean-up the reference applies before "[START]" + text + "[STOP]" (text_utils.py:23-58, tts.py:435)
_PUNC_MAP = (("...", ", "), ("\u2026", ", "), (":", ","), (" - ", ", "), (";", ", "), ("\u2014", "-"), ("\u2013", "-"), (" ,", ","),
             ("\u201c", '"'), ("\u201d", '"'), ("\u2018", "'"), ("\u2019", "'"))
_SENTENCE_ENDERS = (".", "!", "?", "-", ",", "\u3001", "\uff0c", "\u3002", "\uff1f", "\uff01")
_EMPTY_TEXT = "You need to add some text for me to talk."


def punc_norm(text: str) -> str:
    """Capitalise the first letter, collapse whitespace, map rare punctuation, make sure the text ends a sentence."""
    if not text:
        return _EMPTY_TEXT
    if text[0].islower():
        text = text[0].upper() + text[1:]
    text = " ".join(text.split())
    for src, dst in _PUNC_MAP:                 # order matters: "..." before ":" etc.
        text = text.replace(src, dst)
    text = text.rstrip(" ")
    return text if text.endswith(_SENTENCE_ENDERS) else text + "."


def build_prompt_strings(texts: Sequence[str], language_id: Optional[str] = None, multilingual: bool = False) -> List[str]:
    """The strings `ChatterboxTTS.generate_with_conds` hands to `LLM.generate` (tts.py:435-441)."""
    prompts = ["[START]" + punc_norm(t) + "[STOP]" for t in texts]
    if multilingual:
        if not language_id:
            raise ValueError("language_id is required for the multilingual model")
        prompts = [f"<{language_id.lower()}>{p}" for p in prompts]
    return prompts


def _korean_normalize(text: str) -> str:
    """Hangul syllables -> conjoining jamo (initial 0x1100+, medial 0x1161+, final 0x11A7+), then strip."""
    out = []
    for ch in text:
        if "\uac00" <= ch <= "\ud7af":
            base = ord(ch) - 0xAC00
            out.append(chr(0x1100 + base // 588) + chr(0x1161 + (base % 588) // 28) + (chr(0x11A7 + base % 28) if base % 28 else ""))
        else:
            out.append(ch)
    return "".join(out).strip()


class TextTokenizer:
    """kind = "EnTokenizer" | "MtlTokenizer" (the names the reference registers, t3/__init__.py:6-7)."""

    def __init__(self, kind: str, path: str, strict: bool = True):
        """strict: refuse the four languages whose reference normaliser cannot run here (zh / ja / he / ru) instead of
        emitting un-normalised ids; strict=False is for fixtures that want the raw ids (tests/golden/make_golden.py g9)."""
        from tokenizers import Tokenizer

        if kind not in _FILES:
            raise ValueError(f"unknown tokenizer {kind!r}")
        self.kind = kind
        self.strict = strict
        self.tok = Tokenizer.from_file(path)

    @property
    def vocab_size(self) -> int:
        return self.tok.get_vocab_size()

    def encode(self, text: str) -> List[int]:
        if self.kind == "EnTokenizer":
            return self.tok.encode(text.replace(" ", SPACE)).ids          # entokenizer.py:67-69
        language_id = None
        if text.startswith("<"):                                             # mtltokenizer.py:303-306
            language_id = text.split("<")[1].split(">")[0]
            text = text.split(">")[1]
        text = normalize("NFKD", text.lower())                               # mtltokenizer.py:284-298
        if language_id == "ko":
            text = _korean_normalize(text)                                   # mtltokenizer.py:106-125, 317-318
        elif language_id in ("zh", "ja", "he", "ru"):
            # the reference runs Cangjie / kakasi / dicta / a stress marker here (mtltokenizer.py:311-320); those need
            # packages or a download that this image does not have, so these four languages are not pinned.
            if self.strict:
                raise ValueError(f"language {language_id!r} needs a text normaliser (mtltokenizer.py:311-320) that is not available in this "
                                 "build: its ids would differ from the reference's; pass prompt_token_ids instead")
        if language_id:
            text = f"[{language_id.lower()}]{text}"
        return self.tok.encode(text.replace(" ", SPACE)).ids
