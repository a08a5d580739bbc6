"""Conditioning encoder on the device: host-side mirror of the reference's ``T3CondEnc``
(models/t3/modules/cond_enc.py:57-123) over the C ABI of include/t3_engine.h (``t3_cond_*``).

    enc = T3CondEnc(); enc.load_state_dict(state_dict_of_cond_enc)
    cond_emb = enc(speaker_emb, cond_prompt_speech_emb, emotion_adv=0.5)        # [34, 1024] fp32 CPU, as tts.py:279-284 returns
    cond_emb = enc.update_exaggeration(cond_emb, 0.7)                           # tts.py:287-298

All arithmetic runs in the HIP kernels of csrc/cond_enc.hip; there is no CPU path.
"""
from __future__ import annotations

import ctypes as ct
import os
from typing import Iterable, Mapping, Tuple, Union

import torch

from . import constants as C
from .engine import T3Error, T3_E_INVALID, T3_E_NOTFOUND, load_library

PARAM_NAMES = ("spkr_enc.weight", "spkr_enc.bias", "emotion_adv_fc.weight", "perceiver.pre_attention_query",
               "perceiver.attn.norm.weight", "perceiver.attn.norm.bias",
               "perceiver.attn.to_q.weight", "perceiver.attn.to_q.bias", "perceiver.attn.to_k.weight", "perceiver.attn.to_k.bias",
               "perceiver.attn.to_v.weight", "perceiver.attn.to_v.bias", "perceiver.attn.proj_out.weight", "perceiver.attn.proj_out.bias")


class T3CondEnc:
    def __init__(self, device_id: int | None = None):
        self.lib = load_library()
        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.device_count() > 1 else 0
        self.h = ct.c_void_p()
        rc = self.lib.t3_cond_create(int(device_id), ct.byref(self.h))
        if rc:
            self.h = None
            self._raise(rc, self.lib.t3_cond_last_error(None).decode())

    @staticmethod
    def _raise(rc: int, msg: str):
        if rc == T3_E_INVALID:
            raise ValueError(msg)
        raise T3Error(f"[{rc}] {msg}")

    def _chk(self, rc: int):
        if rc:
            self._raise(rc, self.lib.t3_cond_last_error(self.h).decode())

    def load_state_dict(self, tensors: Union[Mapping[str, torch.Tensor], Iterable[Tuple[str, torch.Tensor]]], strict: bool = True):
        """Names as in the reference checkpoint (``cond_enc.*`` inside t3_cfg.safetensors) or as in ``T3CondEnc.state_dict()``.
        Tensors that do not belong to the encoder are skipped; with ``strict`` every encoder tensor must be present."""
        items = tensors.items() if isinstance(tensors, Mapping) else tensors
        seen = set()
        for name, t in items:
            short = name[len("cond_enc."):] if name.startswith("cond_enc.") else name
            if short not in PARAM_NAMES:
                continue
            t = t.detach().to("cpu", torch.float32).contiguous()
            self._chk(self.lib.t3_cond_load_tensor(self.h, name.encode(), t.data_ptr(), t.numel()))
            seen.add(short)
        missing = [n for n in PARAM_NAMES if n not in seen]
        if strict and missing:
            raise KeyError(f"missing conditioning-encoder tensors: {missing}")
        return missing

    def __call__(self, speaker_emb: torch.Tensor, cond_prompt_speech_emb: torch.Tensor, emotion_adv: float = 0.5) -> torch.Tensor:
        spk = speaker_emb.detach().to("cpu", torch.float32).reshape(-1).contiguous()
        if spk.numel() != 256:
            raise ValueError("speaker_emb must have 256 elements (t3_config.py speaker_embed_size)")
        pe = cond_prompt_speech_emb.detach().to("cpu", torch.float32)
        if pe.dim() == 3:                                    # T3Cond drops a leading batch dimension (cond_enc.py:28-30)
            pe = pe[0]
        if pe.dim() != 2 or pe.shape[1] != C.HIDDEN:
            raise ValueError("cond_prompt_speech_emb must be [n, 1024]")
        pe = pe.contiguous()
        out = torch.empty(C.CONDITIONING_SIZE, C.HIDDEN, dtype=torch.float32)
        self._chk(self.lib.t3_cond_encode(self.h, spk.data_ptr(), pe.data_ptr(), pe.shape[0], ct.c_float(float(emotion_adv)), out.data_ptr()))
        return out

    forward = __call__

    def emotion_adv_fc(self, exaggeration: float) -> torch.Tensor:
        out = torch.empty(1, C.HIDDEN, dtype=torch.float32)
        self._chk(self.lib.t3_cond_emotion_row(self.h, ct.c_float(float(exaggeration)), out.data_ptr()))
        return out

    def update_exaggeration(self, cond_emb: torch.Tensor, exaggeration: float) -> torch.Tensor:
        """tts.py:287-298: unchanged at 0.5, otherwise a copy with the last row replaced."""
        if exaggeration == 0.5:
            return cond_emb
        new = cond_emb.clone()
        new[-1] = self.emotion_adv_fc(exaggeration)[0].to(new.dtype)
        return new

    def close(self):
        if getattr(self, "h", None):
            self.lib.t3_cond_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
