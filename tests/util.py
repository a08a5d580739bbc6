"""Shared helpers for the parity tests."""
import numpy as np
import torch

from chatterbox_vllm2_amd.prompt import assemble_prompt_ids


def rand_bf16(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def bits(t: torch.Tensor) -> torch.Tensor:
    """Bit pattern view for exact comparison (NaN-safe, distinguishes -0/+0)."""
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16)
    if t.dtype == torch.float32:
        return t.contiguous().view(torch.int32)
    return t


def assert_bit_equal(a: torch.Tensor, b: torch.Tensor, what=""):
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ne = bits(a) != bits(b)
    if ne.any():
        idx = ne.nonzero()[0].tolist()
        raise AssertionError(f"{what}: {int(ne.sum())}/{ne.numel()} elements differ; first at {idx}: "
                             f"{a[tuple(idx)].item()!r} vs {b[tuple(idx)].item()!r}")


def make_prompt(n_text: int, seed: int, vocab: int = 704, first: int = 255):
    rs = np.random.RandomState(seed)
    text = [first] + rs.randint(0, 695, size=n_text - 2).tolist() + [0]
    return assemble_prompt_ids(text)
