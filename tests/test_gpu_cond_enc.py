"""SURVEY.md 8 f3 -- conditioning encoder on the device, through the C ABI.
Parity bars: kernels and the whole encoder BIT-EXACT against the oracle (0 ulp: the contract fixes every summation order);
the encoder against the committed outputs of the reference's own T3CondEnc (tests/golden/cond_enc_synth.npz, fp32 torch CPU,
unknown summation order) within 2e-5 absolute on values of magnitude ~0.5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TOL_VS_REFERENCE = 2e-5


@pytest.fixture(scope="module")
def E():
    from chatterbox_vllm2_amd import engine
    engine.load_library()
    return engine


@pytest.fixture(scope="module")
def params():
    from chatterbox_vllm2_amd.weights import synthetic_cond_enc_tensors
    return dict(synthetic_cond_enc_tensors(4321))


def bit_equal(a, b, what):
    assert a.shape == b.shape, what
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"{what}: max abs diff {(a - b).abs().max().item():.3e}"


def test_layernorm_linear_attention_kernels(E, oracle, params):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 1024, generator=g) * 2 + 0.3
    w, b = params["cond_enc.perceiver.attn.norm.weight"], params["cond_enc.perceiver.attn.norm.bias"]
    bit_equal(E.k_ce_layernorm(x, w, b), oracle.ce_layernorm(x, w, b), "layernorm")
    W, bias = params["cond_enc.perceiver.attn.to_k.weight"], params["cond_enc.perceiver.attn.to_k.bias"]
    for M in (1, 5, 16, 37):                                  # ragged row counts around the 16-row tile
        xs = x[:M]
        bit_equal(E.k_ce_linear(xs, W, bias), oracle.ce_linear(xs, W, bias), f"linear M={M}")
    r = torch.randn(37, 1024, generator=g)
    bit_equal(E.k_ce_linear(x, W, bias, r), oracle.ce_linear(x, W, bias, r), "linear + residual")
    Ws = params["cond_enc.spkr_enc.weight"]                   # K = 256, M = 1, no bias
    spk = torch.randn(1, 256, generator=g)
    bit_equal(E.k_ce_linear(spk, Ws), oracle.ce_linear(spk, Ws), "speaker projection")
    q = torch.randn(32, 1024, generator=g)
    for nk in (1, 32, 64, 65, 150, 192):                      # key counts around the 64-lane ownership boundaries
        k = torch.randn(nk, 1024, generator=g); v = torch.randn(nk, 1024, generator=g)
        bit_equal(E.k_ce_attention(q, k, v), oracle.ce_attention(q, k, v), f"attention nk={nk}")
    with pytest.raises(ValueError):
        E.k_ce_attention(q, torch.zeros(193, 1024), torch.zeros(193, 1024))


@pytest.mark.parametrize("n", [150, 37])
def test_encoder_matches_oracle_bitwise_and_reference_golden(E, oracle, params, n):
    from chatterbox_vllm2_amd.cond_enc import T3CondEnc
    from chatterbox_vllm2_amd.weights import synthetic_cond_inputs
    enc = T3CondEnc()
    assert enc.load_state_dict(params) == []
    spk, prompt, emo = synthetic_cond_inputs(7, n)
    got = enc(spk, prompt, emo)
    assert got.shape == (34, 1024)
    bit_equal(got, oracle.cond_enc(params, spk, prompt, emo), "encoder vs oracle")
    ref = torch.from_numpy(np.load(os.path.join(G, "cond_enc_synth.npz"))[f"cond_emb_n{n}"])
    assert (got - ref).abs().max().item() < TOL_VS_REFERENCE
    # exaggeration row (tts.py:287-298)
    row = torch.from_numpy(np.load(os.path.join(G, "cond_enc_synth.npz"))["emotion_row_0p9"])
    assert torch.equal(enc.emotion_adv_fc(0.9), row)          # a single product per element: exact
    upd = enc.update_exaggeration(got, 0.9)
    assert torch.equal(upd[:-1], got[:-1]) and torch.equal(upd[-1], row[0]) and enc.update_exaggeration(got, 0.5) is got
    enc.close()


def test_encoder_errors(E, params):
    from chatterbox_vllm2_amd.cond_enc import T3CondEnc
    enc = T3CondEnc()
    with pytest.raises(KeyError):
        enc.load_state_dict({k: v for k, v in params.items() if "to_v" not in k})
    with pytest.raises(Exception):                            # encode before all tensors are there
        enc(torch.zeros(256), torch.zeros(10, 1024))
    enc.load_state_dict(params)
    with pytest.raises(ValueError):
        enc(torch.zeros(255), torch.zeros(10, 1024))
    with pytest.raises(ValueError):
        enc(torch.zeros(256), torch.zeros(0, 1024))
    with pytest.raises(ValueError):
        enc(torch.zeros(256), torch.zeros(193, 1024))
    enc.close()


def test_encoder_output_drives_the_engine(E, params):
    """End of the widened path: raw (speaker, prompt embedding, emotion) -> cond_emb -> T3 decode, all on the device."""
    from chatterbox_vllm2_amd.cond_enc import T3CondEnc
    from chatterbox_vllm2_amd.weights import synthetic_cond_inputs, synthetic_tensors
    from util import make_prompt
    enc = T3CondEnc(); enc.load_state_dict(params)
    cond = enc(*synthetic_cond_inputs(7, 150))
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=200, max_seqs=2, kv_bytes=1 << 28)
    eng.load_tensors(synthetic_tensors(2, 704, 1234)); eng.finalize()
    eng.add_request(0, make_prompt(10, seed=3), cond, E.make_sampling(temperature=0.0, max_tokens=8, ignore_eos=True))
    eng.run_until_done()
    ids, reason = eng.get_output(0)
    assert len(ids) == 8 and reason == 2 and all(2500 <= t < 2500 + 8194 for t in ids)
    eng.close(); enc.close()
