"""Weight sources for the T3 engine.

* ``synthetic_tensors`` -- seeded random weights with the real shapes (no checkpoint is available
  offline; BASELINE.md section 2: seed 1234, N(0, 0.02^2) matrices, RMSNorm weights 1).  This is what
  ``LLM(..., load_format="dummy")`` uses.
* ``iter_safetensors`` -- the reference checkpoint layout (tensor names as routed by
  ``T3VllmModel.load_weights``, reference t3.py:300-332: ``tfmr.*`` -> Llama, else first path
  component; unknown prefixes are skipped, t3.py:316-319).
"""
from __future__ import annotations

import os
from typing import Iterator, Tuple

import torch

from . import constants as C


def synthetic_tensors(n_layers: int = C.N_LAYERS, text_vocab: int = C.TEXT_VOCAB_EN,
                      seed: int = 1234, std: float = 0.02) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield (checkpoint name, bf16 CPU tensor) in a fixed order from one seeded generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def mat(r, c):
        return (torch.randn(r, c, generator=g, dtype=torch.float32) * std).to(torch.bfloat16)

    ones = torch.ones(C.HIDDEN, dtype=torch.bfloat16)
    yield "text_emb.weight", mat(text_vocab, C.HIDDEN)
    yield "speech_emb.weight", mat(C.SPEECH_VOCAB, C.HIDDEN)
    yield "text_pos_emb.emb.weight", mat(C.MAX_TEXT_POS, C.HIDDEN)
    yield "speech_pos_emb.emb.weight", mat(C.MAX_SPEECH_POS, C.HIDDEN)
    yield "speech_head.weight", mat(C.SPEECH_VOCAB, C.HIDDEN)
    yield "tfmr.norm.weight", ones.clone()
    for i in range(n_layers):
        p = f"tfmr.layers.{i}."
        yield p + "self_attn.q_proj.weight", mat(C.HIDDEN, C.HIDDEN)
        yield p + "self_attn.k_proj.weight", mat(C.HIDDEN, C.HIDDEN)
        yield p + "self_attn.v_proj.weight", mat(C.HIDDEN, C.HIDDEN)
        yield p + "self_attn.o_proj.weight", mat(C.HIDDEN, C.HIDDEN)
        yield p + "mlp.gate_proj.weight", mat(C.FFN, C.HIDDEN)
        yield p + "mlp.up_proj.weight", mat(C.FFN, C.HIDDEN)
        yield p + "mlp.down_proj.weight", mat(C.HIDDEN, C.FFN)
        yield p + "input_layernorm.weight", ones.clone()
        yield p + "post_attention_layernorm.weight", ones.clone()


def synthetic_cond_emb(seed: int = 1) -> torch.Tensor:
    """[34, 1024] fp32 conditioning stand-in (N(0,1)*0.02, seed 1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(C.CONDITIONING_SIZE, C.HIDDEN, generator=g, dtype=torch.float32) * 0.02


COND_ENC_PARAMS = (                      # checkpoint name under "cond_enc." -> shape   (cond_enc.py:62-78, perceiver.py:137-148, 190-201)
    ("spkr_enc.weight", (C.HIDDEN, 256)), ("spkr_enc.bias", (C.HIDDEN,)), ("emotion_adv_fc.weight", (C.HIDDEN, 1)),
    ("perceiver.pre_attention_query", (1, 32, C.HIDDEN)),
    ("perceiver.attn.norm.weight", (C.HIDDEN,)), ("perceiver.attn.norm.bias", (C.HIDDEN,)),
    ("perceiver.attn.to_q.weight", (C.HIDDEN, C.HIDDEN)), ("perceiver.attn.to_q.bias", (C.HIDDEN,)),
    ("perceiver.attn.to_k.weight", (C.HIDDEN, C.HIDDEN)), ("perceiver.attn.to_k.bias", (C.HIDDEN,)),
    ("perceiver.attn.to_v.weight", (C.HIDDEN, C.HIDDEN)), ("perceiver.attn.to_v.bias", (C.HIDDEN,)),
    ("perceiver.attn.proj_out.weight", (C.HIDDEN, C.HIDDEN)), ("perceiver.attn.proj_out.bias", (C.HIDDEN,)),
)


def synthetic_cond_enc_tensors(seed: int = 4321) -> Iterator[Tuple[str, torch.Tensor]]:
    """Seeded fp32 parameters of the conditioning encoder (4 497 408 values), checkpoint names and shapes of the reference's
    T3CondEnc.  Magnitudes chosen so that every stage matters: O(1/sqrt(fan_in)) matrices, LayerNorm weight near 1."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name, shape in COND_ENC_PARAMS:
        if name.endswith("norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("pre_attention_query"):
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.3
        else:
            fan_in = shape[-1]
            t = torch.randn(shape, generator=g) / (fan_in ** 0.5)
        yield "cond_enc." + name, t.to(torch.float32).contiguous()


def synthetic_cond_inputs(seed: int = 7, n_prompt: int = 150):
    """(speaker_emb [1,256], cond_prompt_speech_emb [n,1024], emotion_adv) stand-ins for tts.py:277-284."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    spk = torch.randn(1, 256, generator=g) * 0.1
    prompt = torch.randn(n_prompt, C.HIDDEN, generator=g) * 0.05
    return spk, prompt, 0.5


def iter_safetensors(path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield (name, bf16 tensor) from a reference checkpoint (t3_cfg.safetensors / t3_mtl23ls_v2.safetensors,
    or the ``model.safetensors`` symlink the reference creates, tts.py:225-229)."""
    from safetensors import safe_open

    if os.path.isdir(path):
        path = os.path.join(path, "model.safetensors")
    with safe_open(path, framework="pt", device="cpu") as f:
        for name in f.keys():
            yield name, f.get_tensor(name).to(torch.bfloat16)
