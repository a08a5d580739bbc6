"""The independent fidelity gate of the oracle, shared by tests/test_oracle.py (runs it live) and
tests/golden/make_golden.py g4 (runs it at full length and commits the HF vectors + the per-step evidence as tests/golden/hf_gate.npz).

What is compared: the oracle's whole decode path -- both CFG streams, all layers, KV-cache decode with the speech-position add
(t3.py:440-480), final norm + speech head + CFG (t3.py:650-662) -- against transformers' LlamaModel on the same synthetic weights,
(a) in fp32 (the arithmetic both are rounding) and (b) in bf16 (the dtype the reference's vLLM runs in), all three teacher-forced
with the SAME ids (the oracle's greedy stream), every RMSNorm given a non-trivial weight so that the load-time fold is exercised.
The oracle must be as close to HF-fp32 as HF's own bf16 run is."""
import numpy as np
import torch

NORM_SEED = 77


def gate_tensors(n_layers, vocab):
    """The synthetic checkpoint with every RMSNorm weight replaced by 1 + 0.25 N(0, 1) (seeded): the shipped one has unit norm weights."""
    from chatterbox_vllm2_amd.weights import synthetic_tensors
    g = torch.Generator().manual_seed(NORM_SEED)
    return [(k, (1.0 + 0.25 * torch.randn(v.shape, generator=g)).to(torch.bfloat16) if "norm" in k else v) for k, v in synthetic_tensors(n_layers, vocab, 1234)]


def hf_model(tensors, n_layers, dtype):
    from transformers import LlamaConfig, LlamaModel
    cfg = LlamaConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=n_layers, num_attention_heads=16,
                      num_key_value_heads=16, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0, vocab_size=8,
                      rope_scaling={"factor": 8.0, "high_freq_factor": 4.0, "low_freq_factor": 1.0,
                                    "original_max_position_embeddings": 8192, "rope_type": "llama3"},
                      max_position_embeddings=131072, attention_bias=False, mlp_bias=False, hidden_act="silu",
                      attn_implementation="eager")
    hf = LlamaModel(cfg).eval()
    sd = {k[5:]: v.float() for k, v in tensors if k.startswith("tfmr.")}
    sd["embed_tokens.weight"] = torch.zeros(8, 1024)
    hf.load_state_dict(sd)
    return hf.to(dtype)


def nucleus(logits: torch.Tensor, temperature=0.8, top_p=0.8) -> torch.Tensor:
    """Boolean mask of the tokens vLLM's top-p keeps (sort ascending, drop the prefix whose cumulative mass is <= 1 - top_p, never the
    last) at the reference's sampling defaults (tts.py:377, api_server.py:45): the set that decides which ids can be SAMPLED."""
    p = torch.softmax(logits.double() / temperature, dim=-1)
    sp, idx = torch.sort(p, dim=-1, descending=False)
    drop = torch.cumsum(sp, dim=-1) <= (1.0 - top_p)
    drop[..., -1] = False
    keep = torch.zeros_like(drop)
    keep.scatter_(-1, idx, ~drop)
    return keep


def oracle_teacher_forced(oracle, tens, n_layers, vocab, prompt, cond, n_steps, taps=(), tap_steps=None, want_embeds=False):
    """The oracle's greedy decode driven row by row (prefill rows of both CFG streams, then one row pair per step), so that the
    residual stream can be tapped after any number of layers.  Returns ids, post-CFG logits [n, 8194], the conditional / unconditional
    final residual rows [n, 2, 1024] and the tapped rows {layers: [tap_steps, 2, 1024]} of the row that produces each step's logits
    (tap_steps: only the first so many steps are tapped; a tap costs one more pass over the rows); want_embeds: also (ec, eu)."""
    w = dict(tens)
    T = len(prompt)
    m = oracle.OracleModel(n_layers, vocab, max_pos=T + n_steps + 2).load(tens)
    ec, eu = m.prompt_embeds(prompt, cond)
    rows = torch.cat([ec, eu]); rs = [0] * T + [1] * T; rp = list(range(T)) * 2
    tapped = {k: [] for k in taps}
    finals, ids, logits = [], [], []

    def run(h, rs_, rp_, pick, step):
        out = None
        for k in (taps if tap_steps is None or step < tap_steps else ()):      # one pass per tap (the oracle taps one layer per call); KV writes are idempotent
            _, t = m.forward_rows(h, rs_, rp_, tap_layer=k)
            tapped[k].append(t[pick].clone())
        out, _ = m.forward_rows(h, rs_, rp_)
        return out[pick]

    hcu = run(rows, rs, rp, [T - 1, 2 * T - 1], 0)
    semb, spos = w["speech_emb.weight"].float(), w["speech_pos_emb.emb.weight"].float()
    for k in range(n_steps):
        finals.append(hcu.clone())
        lg = m.cfg_logits(hcu[0], hcu[1], 0.5)
        logits.append(lg)
        tok = int(torch.argmax(lg))                   # first maximum, repetition penalty off: the greedy rule of orc_sample
        ids.append(tok)
        if k == n_steps - 1:
            break
        x = (semb[tok] + spos[k + 1]).to(torch.bfloat16)
        hcu = run(torch.stack([x, x]), [0, 1], [T + k, T + k], [0, 1], k + 1)
    m.close()
    res = (ids, torch.stack(logits), torch.stack(finals), {k: torch.stack(v) for k, v in tapped.items()})
    return res + ((ec, eu),) if want_embeds else res


def hf_teacher_forced(tens, n_layers, ec, eu, ids, dtype, taps=()):
    """transformers' LlamaModel (fp32 or bf16) over the same prompt embeddings and the same teacher-forced ids, KV cache per CFG stream.
    Returns post-CFG logits [n, 8194] (fp32: exact arithmetic of t3.py:662; bf16: bf16 tensor arithmetic as the reference runs it),
    the post-final-norm hidden rows [n, 2, 1024] and hidden states after `taps` layers [n, 2, 1024] (taps < n_layers)."""
    from transformers import DynamicCache
    w = dict(tens)
    # one-token steps are GEMV-sized: more than a few OpenMP threads only add fork / join cost (and thrash badly on a shared box)
    n_thr = torch.get_num_threads(); torch.set_num_threads(min(4, n_thr))
    hf = hf_model(tens, n_layers, dtype)
    head = w["speech_head.weight"].to(dtype)
    semb, spos = w["speech_emb.weight"].float(), w["speech_pos_emb.emb.weight"].float()
    caches = [DynamicCache(config=hf.config), DynamicCache(config=hf.config)]
    logits, hidden, tapped = [], [], {k: [] for k in taps}

    def step(inputs):
        outs = [hf(inputs_embeds=x.to(dtype), past_key_values=caches[s], use_cache=True, output_hidden_states=bool(taps)) for s, x in enumerate(inputs)]
        for k in taps:
            tapped[k].append(torch.stack([o.hidden_states[k][0, -1].float() for o in outs]))
        return [o.last_hidden_state[0, -1] for o in outs]

    with torch.no_grad():
        hs = step([ec[None], eu[None]])
        for k in range(len(ids)):
            hidden.append(torch.stack([h.float() for h in hs]))
            lc, lu = hs[0] @ head.T, hs[1] @ head.T
            logits.append((lc + 0.5 * (lc - lu)).float())                          # t3.py:662, in the model dtype
            if k == len(ids) - 1:
                break
            x = (semb[ids[k]] + spos[k + 1]).to(torch.bfloat16)[None, None]        # decode embedding: a bf16 tensor in every run
            hs = step([x, x])
    torch.set_num_threads(n_thr)
    return torch.stack(logits), torch.stack(hidden), {k: torch.stack(v) for k, v in tapped.items()}


def compare(lg_oracle, lg_hf32, lg_hfbf16, ids):
    """Per-step figures of merit.  err_* : max and mean |logit - HF fp32 logit|; agree: greedy id == HF fp32 argmax; margin: HF fp32
    top-1 minus top-2; nucleus_*: Jaccard overlap of the top-p = 0.8 / T = 0.8 nucleus with HF fp32's, and the HF-fp32 probability
    mass of the tokens on which the two nuclei differ."""
    out = {}
    for name, lg in (("oracle", lg_oracle), ("hfbf16", lg_hfbf16)):
        if lg is None:
            continue
        d = (lg - lg_hf32).abs()
        out[f"err_max_{name}"] = d.max(dim=1).values.numpy(); out[f"err_mean_{name}"] = d.mean(dim=1).numpy()
        a, b = nucleus(lg), nucleus(lg_hf32)
        p32 = torch.softmax(lg_hf32.double() / 0.8, dim=-1)
        out[f"nucleus_jaccard_{name}"] = ((a & b).sum(1).double() / (a | b).sum(1).double()).numpy()
        out[f"nucleus_diff_mass_{name}"] = (p32 * (a ^ b)).sum(1).numpy()
    top2 = lg_hf32.topk(2, dim=1)
    out["hf32_margin"] = (top2.values[:, 0] - top2.values[:, 1]).numpy()
    out["hf32_top1"] = top2.indices[:, 0].numpy()
    out["agree_oracle"] = (top2.indices[:, 0] == torch.tensor(ids)).numpy()
    out["hf32_nucleus_size"] = nucleus(lg_hf32).sum(1).numpy()
    out["hf32_logit_std"] = np.float64(lg_hf32.std())
    return out
