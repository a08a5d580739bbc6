"""Generates the committed golden fixtures under tests/golden/ (run in the BUILD container, where
/root/reference is mounted; the GPU box only sees the resulting .npz / .json files).

    python tests/golden/make_golden.py

What comes from where:
  G1  cond_enc.npz     -- output of the REFERENCE's own T3CondEnc (+Perceiver, LearnedPositionEmbeddings),
                          imported from /root/reference with stub parent packages (the package
                          __init__ files import vllm, which is not installed), seeded weights + inputs.
  G2  prompt_embeds.npz -- outputs of the reference's own models/t3/t3.py: create_triangular_matrix, split_prefill_decode and the
                          prefill branches of get_input_embeddings (full block + the three chunked variants), and what its decode
                          branch literally returns; t3.py imported with raise-on-use placeholders for the vllm names it imports.
  G3  rope.npz         -- inv_freq and cos/sin of transformers' LlamaRotaryEmbedding for the reference's
                          t3-model/config.json rope settings.
  G4  hf_gate.npz      -- transformers LlamaModel (fp32 and bf16) on the gate's weights: HF-fp32 CFG logits and hidden states after
                          1 / 2 / 30 layers at three steps, plus 128 steps of per-step error / agreement / nucleus-overlap figures
                          of the oracle and of HF-bf16 against HF-fp32 (tests/hf_gate.py).
  G5/G6 streams.npz    -- oracle token streams + post-CFG logits on seeded synthetic weights
                          (2-layer English, 2-layer multilingual batch, 30-layer short) -- regression
                          pins for both the oracle and the GPU engine.
  G8  postfilter.json  -- decisions of the reference's AlignmentStreamAnalyzer (imported, CPU) over random / planted token lists
  G7  tokenizer.json   -- token ids of the fixed en/es utterances from the reference's
                          tokenizer JSON files via the `tokenizers` library.
  G9  c4_requests.json -- token ids of the sentences of the reference's docs/benchmark-text-*.txt (C4 request stream)
  G6b streams30.npz    -- 30-layer oracle streams for C3 uids 0 / 17 and six C4 requests (multilingual vocabulary)
  G6c streams30_full.npz -- FULL-LENGTH 30-layer oracle streams: C3 uids 0 / 17 (884 / 859 tokens), C2 (292), two C4 requests
No reference source text is stored: only inputs and numeric outputs.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
REF_PKG = os.path.join(REF, "src", "chatterbox_vllm")

EN_TEXT = "The quick brown fox jumps over the lazy dog near the river bank while seven birds sing sweetly in tall trees."
ES_TEXT = "El rápido zorro marrón salta sobre el perro perezoso cerca del río mientras siete pájaros cantan dulcemente en los árboles altos."


def import_reference_leaf_modules():
    """Stub the parent packages so that the vllm-importing __init__ files are bypassed."""
    import importlib
    for name, path in [("chatterbox_vllm", REF_PKG), ("chatterbox_vllm.models", REF_PKG + "/models"),
                       ("chatterbox_vllm.models.t3", REF_PKG + "/models/t3"),
                       ("chatterbox_vllm.models.t3.modules", REF_PKG + "/models/t3/modules")]:
        m = types.ModuleType(name); m.__path__ = [path]; sys.modules[name] = m
    cfg = importlib.import_module("chatterbox_vllm.models.t3.modules.t3_config")
    lpe = importlib.import_module("chatterbox_vllm.models.t3.modules.learned_pos_emb")
    ce = importlib.import_module("chatterbox_vllm.models.t3.modules.cond_enc")
    return cfg, lpe, ce


def g1_cond_enc():
    cfg, lpe, ce = import_reference_leaf_modules()
    torch.manual_seed(42)
    hp = cfg.T3Config.multilingual()
    enc = ce.T3CondEnc(hp).eval()
    speech_emb = torch.nn.Embedding(hp.speech_tokens_dict_size, hp.n_channels)
    pos = lpe.LearnedPositionEmbeddings(hp.max_speech_tokens + 2 + 2, hp.n_channels)
    spk = torch.randn(1, hp.speaker_embed_size)
    toks = torch.randint(0, 6561, (1, hp.speech_cond_prompt_len))
    with torch.no_grad():
        prompt_emb = speech_emb(toks)[0] + pos(toks)           # tts.py:277
        out = enc(ce.T3Cond(speaker_emb=spk, cond_prompt_speech_tokens=toks, cond_prompt_speech_emb=prompt_emb,
                            emotion_adv=0.5 * torch.ones(1, 1)))
        out_exag = enc.emotion_adv_fc(0.9 * torch.ones(1, 1))   # tts.py:294-296
        fixed = pos.get_fixed_embedding(torch.tensor([0, 1, 7]))
    assert tuple(out.shape) == (34, 1024)
    np.savez_compressed(os.path.join(HERE, "cond_enc.npz"), cond_emb=out.numpy(), emotion_row_09=out_exag.numpy(),
                        pos_rows_0_1_7=fixed.numpy(), pos_table_rows=pos.emb.weight[[0, 1, 7]].detach().numpy(),
                        n_params=np.int64(sum(p.numel() for p in enc.parameters())),
                        constants=np.array([hp.start_speech_token, hp.stop_speech_token, hp.speech_tokens_dict_size,
                                            hp.max_text_tokens, hp.max_speech_tokens, hp.speech_cond_prompt_len,
                                            hp.n_channels, hp.text_tokens_dict_size, cfg.T3Config.english_only().text_tokens_dict_size]))
    print("G1 cond_enc:", out.shape, float(out.abs().mean()))


def g1b_cond_enc_synthetic():
    """f3 pin: the reference's T3CondEnc (imported) with the product's seeded synthetic parameters loaded into it, fp32 CPU,
    on seeded inputs -> cond_emb [34,1024] for two prompt lengths, plus the exaggeration row of tts.py:287-298."""
    from chatterbox_vllm2_amd.weights import synthetic_cond_enc_tensors, synthetic_cond_inputs
    cfg, lpe, ce = import_reference_leaf_modules()
    hp = cfg.T3Config.multilingual()
    enc = ce.T3CondEnc(hp).eval()
    sd = {k[len("cond_enc."):]: v for k, v in synthetic_cond_enc_tensors(4321)}
    missing, unexpected = enc.load_state_dict(sd, strict=True)
    out = {}
    with torch.no_grad():
        for n in (150, 37):
            spk, prompt, emo = synthetic_cond_inputs(7, n)
            y = enc(ce.T3Cond(speaker_emb=spk, cond_prompt_speech_tokens=torch.zeros(1, n, dtype=torch.long),
                              cond_prompt_speech_emb=prompt, emotion_adv=emo * torch.ones(1, 1)))
            assert tuple(y.shape) == (34, 1024)
            out[f"cond_emb_n{n}"] = y.numpy()
        out["emotion_row_0p9"] = enc.emotion_adv_fc(0.9 * torch.ones(1, 1)).numpy()
    np.savez_compressed(os.path.join(HERE, "cond_enc_synth.npz"), **out)
    print("G1b cond_enc (synthetic params):", {k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})


def g3_rope():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    c = json.load(open(os.path.join(REF, "t3-model", "config.json")))
    cfg = LlamaConfig(hidden_size=1024, num_attention_heads=c["num_attention_heads"], head_dim=c["head_dim"],
                      rope_theta=c["rope_theta"], rope_scaling=c["rope_scaling"], max_position_embeddings=c["max_position_embeddings"])
    rot = LlamaRotaryEmbedding(cfg)
    pos = torch.tensor([[0, 1, 107, 999]])
    cos, sin = rot(torch.zeros(1, 4, 1024), pos)
    np.savez_compressed(os.path.join(HERE, "rope.npz"), inv_freq=rot.inv_freq.numpy(), positions=pos.numpy()[0],
                        cos=cos[0, :, :32].numpy(), sin=sin[0, :, :32].numpy(),
                        llama_cfg=np.array([c["num_hidden_layers"], c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"],
                                            c["intermediate_size"], c["vocab_size"]]), rms_eps=np.float64(c["rms_norm_eps"]))
    print("G3 rope inv_freq[15:18]:", rot.inv_freq[15:18].tolist())


def g7_tokenizer():
    from chatterbox_vllm2_amd.prompt import TextTokenizer
    en = TextTokenizer("EnTokenizer", os.path.join(REF_PKG, "models/t3/tokenizer.json"))
    mtl = TextTokenizer("MtlTokenizer", os.path.join(REF_PKG, "models/t3/grapheme_mtl_merged_expanded_v1.json"))
    out = {
        "en_text": EN_TEXT, "es_text": ES_TEXT,
        "en_english_ids": en.encode("[START]" + EN_TEXT + "[STOP]"),                 # tts.py:435
        "en_mtl_ids": mtl.encode("<en>[START]" + EN_TEXT + "[STOP]"),                # tts.py:441
        "es_mtl_ids": mtl.encode("<es>[START]" + ES_TEXT + "[STOP]"),
        "en_vocab": en.vocab_size, "mtl_vocab": mtl.vocab_size,
        "special": {k: en.tok.token_to_id(k) for k in ("[START]", "[STOP]", "[SPACE]", "[UNK]", "[PLACEHOLDER55]", "[PLACEHOLDER56]", "[PLACEHOLDER57]")},
        "lang": {k: mtl.tok.token_to_id(k) for k in ("[en]", "[es]", "[fr]", "[zh]")},
    }
    json.dump(out, open(os.path.join(HERE, "tokenizer.json"), "w"), indent=1, ensure_ascii=False)
    print("G7 tokenizer: en", len(out["en_english_ids"]), "en/mtl", len(out["en_mtl_ids"]), "es/mtl", len(out["es_mtl_ids"]))


def g7b_tokenizer_cases():
    """ids produced by the reference's OWN tokenizer classes (entokenizer.py / mtltokenizer.py imported as leaf modules,
    their vocabulary JSON files read in place) through PreTrainedTokenizer.encode, the call vLLM makes on a prompt string.
    Languages whose normaliser needs an absent package or a download (zh, ja, he, ru) are not covered."""
    import importlib
    os.environ.setdefault("HF_HUB_OFFLINE", "1")          # the Cangjie table download fails fast; the class tolerates that
    import_reference_leaf_modules()
    en_mod = importlib.import_module("chatterbox_vllm.models.t3.entokenizer")
    mtl_mod = importlib.import_module("chatterbox_vllm.models.t3.mtltokenizer")
    en = en_mod.EnTokenizer.from_pretrained()
    mtl = mtl_mod.MTLTokenizer.from_pretrained()
    en_texts = [EN_TEXT, "Hello, world!", "  two  spaces  and trailing ", "UPPER lower MiXeD 123 4.5%", "Don't stop -- believing; it's 9:30?",
                "naïve café déjà vu", "a", "", "tabs\tand\nnewlines", "quotes \"double\" and 'single' (parens) [brackets]",
                "emoji \U0001F600 and symbols © ™ € £", "x" * 300]
    mtl_texts = [("en", EN_TEXT), ("es", ES_TEXT), ("fr", "Où est la bibliothèque, s'il vous plaît ?"), ("de", "Größe und Straße: Äpfel, Öl, Übung."),
                 ("it", "Perché no? È così!"), ("pt", "Não há ação sem coração."), ("ko", "안녕하세요, 만나서 반갑습니다."),
                 ("en", "a > b is dropped after the second bracket"), ("EN", "Upper-case language tag"), (None, "No language tag at all"),
                 ("en", ""), ("pl", "Zażółć gęślą jaźń"), ("tr", "İstanbul'da ılık bir gün"), ("hi", "नमस्ते दुनिया"), ("ar", "مرحبا بالعالم")]
    out = {"en": [], "mtl": []}
    for t in en_texts:
        prompt = "[START]" + t + "[STOP]"                                   # tts.py:435
        out["en"].append({"prompt": prompt, "ids": [int(i) for i in en.encode(prompt)]})
    for lang, t in mtl_texts:
        prompt = (f"<{lang}>" if lang else "") + "[START]" + t + "[STOP]"    # tts.py:441
        out["mtl"].append({"prompt": prompt, "ids": [int(i) for i in mtl.encode(prompt)]})
    # text clean-up of tts.py:435 (text_utils.punc_norm, imported)
    tu = importlib.import_module("chatterbox_vllm.text_utils")
    raw = ["", " ", "hello world", "Already Capitalised.", "wait... what", "ellipsis… char", "colon: here; semi", "a - b — c – d", "space , comma",
           "“curly” ‘quotes’", "  many   spaces\tand\nlines  ", "ends with dash-", "ends with comma,", "question?", "日本語のテキスト。", "中文，", "trailing space ",
           "éclair au chocolat", "1 2 3", "x"]
    out["punc_norm"] = [{"text": t, "out": tu.punc_norm(t)} for t in raw]
    json.dump(out, open(os.path.join(HERE, "tokenizer_cases.json"), "w"), indent=1, ensure_ascii=False)
    print("G7b tokenizer cases:", len(out["en"]), "en,", len(out["mtl"]), "mtl")


def g6_streams():
    from oracle import oracle as O
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors
    tok = json.load(open(os.path.join(HERE, "tokenizer.json")))
    cond = synthetic_cond_emb(1)
    out = {}
    # (a) 2-layer English, C1 prompt (T = 108): greedy + the reference's sampling defaults
    m = O.OracleModel(2, 704, max_pos=400).load(synthetic_tensors(2, 704, 1234))
    p_en = assemble_prompt_ids(tok["en_english_ids"])
    ids, lg = m.generate(p_en, cond, O.make_sampling(temperature=0.0, max_tokens=64, ignore_eos=True), want_logits=True, max_model_len=400)
    out["l2_en_greedy_ids"] = np.array(ids, np.int32); out["l2_en_greedy_logits_step0"] = lg[0].numpy(); out["l2_en_greedy_logits_step63"] = lg[63].numpy()
    ids, _ = m.generate(p_en, cond, O.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=0, max_tokens=64, ignore_eos=True), max_model_len=400)
    out["l2_en_sampled_ids"] = np.array(ids, np.int32)
    m.close()
    # (b) 2-layer multilingual: en (T=116) and es (T=141), uid 0/1, sampled
    m = O.OracleModel(2, 2454, max_pos=400).load(synthetic_tensors(2, 2454, 1234))
    for name, key, uid in (("en", "en_mtl_ids", 0), ("es", "es_mtl_ids", 1)):
        ids, _ = m.generate(assemble_prompt_ids(tok[key]), cond, O.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=uid, max_tokens=48, ignore_eos=True), max_model_len=400)
        out[f"l2_mtl_{name}_sampled_ids"] = np.array(ids, np.int32)
    m.close()
    # (c) 30-layer English (the real depth), C1 prompt, 16 greedy + 16 sampled tokens
    m = O.OracleModel(30, 704, max_pos=200).load(synthetic_tensors(30, 704, 1234))
    ids, lg = m.generate(p_en, cond, O.make_sampling(temperature=0.0, max_tokens=16, ignore_eos=True), want_logits=True, max_model_len=200)
    out["l30_en_greedy_ids"] = np.array(ids, np.int32); out["l30_en_greedy_logits_step0"] = lg[0].numpy()
    ids, _ = m.generate(p_en, cond, O.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=0, max_tokens=16, ignore_eos=True), max_model_len=200)
    out["l30_en_sampled_ids"] = np.array(ids, np.int32)
    m.close()
    np.savez_compressed(os.path.join(HERE, "streams.npz"), **out)
    print("G6 streams:", {k: v.shape for k, v in out.items()})


def g9_c4_requests():
    """C4 request stream (BASELINE.json configs[3]): sentences of the reference's docs/benchmark-text-{1,2,fr-1,zh-1}.txt,
    split on sentence enders, cleaned with punc_norm, decorated as tts.py:435-441 does and tokenised HERE with the f2 tokenizer over the
    reference's multilingual vocabulary file (zh: raw ids, the Cangjie table is an un-fetchable download).  Only token ids are stored, plus a
    per-request output length G ~ U{200..800} (seed 7) capped so that T + G <= max_model_len = 1000."""
    import re
    from chatterbox_vllm2_amd.prompt import TextTokenizer, assemble_prompt_ids, punc_norm
    mtl = TextTokenizer("MtlTokenizer", os.path.join(REF_PKG, "models/t3/grapheme_mtl_merged_expanded_v1.json"), strict=False)
    rs = np.random.RandomState(7)
    reqs = []
    for fname, lang in (("1", "en"), ("2", "en"), ("fr-1", "fr"), ("zh-1", "zh")):
        raw = open(os.path.join(REF, "docs", f"benchmark-text-{fname}.txt"), encoding="utf-8").read()
        body = " ".join(l for l in raw.splitlines() if not l.startswith("#"))
        sents = [x.strip() for x in re.split(r"(?<=[.!?\u3002\uff01\uff1f])\s*", body) if len(x.strip()) >= 2]
        for k, sent in enumerate(sents):
            ids = mtl.encode(f"<{lang}>[START]{punc_norm(sent)}[STOP]")
            T = len(assemble_prompt_ids(ids))
            g = int(rs.randint(200, 801))
            reqs.append({"src": f"{fname}:{k}", "lang": lang, "text_ids": [int(i) for i in ids], "max_tokens": min(g, 1000 - T - 1)})
    out = {"max_model_len": 1000, "slots": 128, "seed_G": 7, "sampling": {"temperature": 0.8, "top_p": 0.8, "repetition_penalty": 2.0, "seed": 0},
           "uid": "request index", "requests": reqs}
    json.dump(out, open(os.path.join(HERE, "c4_requests.json"), "w"), separators=(",", ":"))
    print("G9 c4 requests:", len(reqs), "total output tokens", sum(r["max_tokens"] for r in reqs),
          "text ids min/mean/max", min(len(r["text_ids"]) for r in reqs), sum(len(r["text_ids"]) for r in reqs) // len(reqs), max(len(r["text_ids"]) for r in reqs))


C4_GOLDEN_REQUESTS = (0, 3, 250, 431, 464, 480)     # en (long), en, en (text 2), fr, zh, zh


def g6b_streams_30_layers_multilingual():
    """30-layer oracle streams at the production batch shapes' inputs: C3 uids 0 (en) and 17 (es), and six C4 requests --
    what the 64-row / 256-row GEMM schedules and the continuous-batching loop must reproduce on the device."""
    from oracle import oracle as O
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors
    tok = json.load(open(os.path.join(HERE, "tokenizer.json")))
    c4 = json.load(open(os.path.join(HERE, "c4_requests.json")))
    cond = synthetic_cond_emb(1)
    m = O.OracleModel(30, 2454, max_pos=400).load(synthetic_tensors(30, 2454, 1234))
    out = {}
    kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, ignore_eos=True)
    for uid, key in ((0, "en_mtl_ids"), (17, "es_mtl_ids")):
        ids, lg = m.generate(assemble_prompt_ids(tok[key]), cond, O.make_sampling(uid=uid, max_tokens=24, **kw), want_logits=True, max_model_len=400)
        out[f"c3_uid{uid}_ids"] = np.array(ids, np.int32); out[f"c3_uid{uid}_logits_step0"] = lg[0].numpy()
        print("  c3 uid", uid, ids[:8], flush=True)
    for i in C4_GOLDEN_REQUESTS:
        r = c4["requests"][i]
        ids, _ = m.generate(assemble_prompt_ids(r["text_ids"]), cond, O.make_sampling(uid=i, max_tokens=32, **kw), max_model_len=400)
        out[f"c4_req{i}_ids"] = np.array(ids, np.int32)
        print("  c4 req", i, r["lang"], len(r["text_ids"]), ids[:8], flush=True)
    m.close()
    np.savez_compressed(os.path.join(HERE, "streams30.npz"), **out)
    print("G6b streams30:", {k: v.shape for k, v in out.items()})


C4_FULL_LENGTH_REQUESTS = (431, 464)                 # fr (T = 268, 317 tokens), zh (T = 57, 767 tokens)


def g6c_full_length_streams():
    """FULL-LENGTH 30-layer oracle streams (ids only): C3 utterances 0 (en, 884 tokens) and 17 (es, 859 tokens) to max_model_len 1000,
    C2's 292 tokens (English vocabulary, max_model_len 400) and two C4 requests to their max_tokens -- so that the device path is
    compared with the oracle to the END of the headline configurations (contexts up to 1000, every 256-token KV block boundary, the
    4-wave attention at >= 3 chunks per wave), not only over their first 24-32 tokens.  ~1 h of CPU on 8 cores; saved stream by stream."""
    from oracle import oracle as O
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors
    tok = json.load(open(os.path.join(HERE, "tokenizer.json")))
    c4 = json.load(open(os.path.join(HERE, "c4_requests.json")))
    cond = synthetic_cond_emb(1)
    path = os.path.join(HERE, "streams30_full.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, ignore_eos=True)

    def run(m, key, prompt, sp, mml):
        if key in out:
            return
        import time
        t0 = time.time()
        ids, _ = m.generate(prompt, cond, sp, max_model_len=mml)
        out[key] = np.array(ids, np.int16 if max(ids) < 32768 else np.int32)
        np.savez_compressed(path, **out)
        print(f"  {key}: {len(ids)} tokens, T = {len(prompt)}, {time.time() - t0:.0f} s", ids[:6], flush=True)

    m = O.OracleModel(30, 704, max_pos=400).load(synthetic_tensors(30, 704, 1234))
    run(m, "c2_en_sampled_ids", assemble_prompt_ids(tok["en_english_ids"]), O.make_sampling(uid=0, max_tokens=400 - 108, **kw), 400)
    m.close()
    m = O.OracleModel(30, 2454, max_pos=1000).load(synthetic_tensors(30, 2454, 1234))
    for i in C4_FULL_LENGTH_REQUESTS:
        r = c4["requests"][i]
        run(m, f"c4_req{i}_ids", assemble_prompt_ids(r["text_ids"]), O.make_sampling(uid=i, max_tokens=r["max_tokens"], **kw), 1000)
    for uid, key in ((17, "es_mtl_ids"), (0, "en_mtl_ids")):
        p = assemble_prompt_ids(tok[key])
        run(m, f"c3_uid{uid}_ids", p, O.make_sampling(uid=uid, max_tokens=1000 - len(p), **kw), 1000)
    m.close()
    print("G6c full-length streams:", {k: v.shape for k, v in out.items()})

# ------------------------------------------------------------------------------------------------------------------------
# G2: fixtures produced by the reference's OWN hot-path code (models/t3/t3.py), not by a restatement of it
# ------------------------------------------------------------------------------------------------------------------------
class _Seal:
    sealed = False          # False while t3.py is being imported; True while a fixture case runs
    touched = []            # placeholder uses seen while sealed (must stay empty)


class _PlaceholderMeta(type):
    """Classes standing in for the vllm names t3.py imports (t3.py:9-31).  They exist so that the module's `import` lines, base-class
    lists, annotations and its registration decorator (t3.py:252-254) evaluate; once sealed, ANY use of one -- a call, an
    instantiation, a subscript, an attribute -- raises, so a fixture case cannot have executed a line that depends on vllm."""
    def _guard(cls, what):
        if _Seal.sealed:
            _Seal.touched.append(f"{cls.__name__}: {what}")
            raise AssertionError(f"fixture case touched the vllm placeholder {cls.__name__} ({what})")

    def __call__(cls, *a, **k):
        if not cls.__dict__.get("_placeholder"):
            # a class of t3.py that merely inherits from a placeholder: instantiating it would run vllm's base-class machinery
            cls._guard("instantiation of a class derived from a placeholder")
            return super().__call__(*a, **k)
        cls._guard("call")
        return lambda obj=None, *aa, **kk: obj          # import time only: `@MULTIMODAL_REGISTRY.register_processor(...)` returns the class unchanged

    def __getitem__(cls, item):
        cls._guard("subscript")
        return cls

    def __getattr__(cls, name):
        if name.startswith("__") or not cls.__dict__.get("_placeholder"):
            raise AttributeError(name)
        cls._guard(f"attribute {name}")
        return _PlaceholderMeta(f"{cls.__name__}.{name}", (), {"_placeholder": True})


class _PlaceholderModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if _Seal.sealed:
            _Seal.touched.append(f"{self.__name__}.{name}")
            raise AssertionError(f"fixture case touched the vllm placeholder module {self.__name__}.{name}")
        ph = _PlaceholderMeta(name, (), {"_placeholder": True})
        setattr(self, name, ph)
        return ph


VLLM_MODULES = ("vllm", "vllm.config", "vllm.model_executor", "vllm.model_executor.layers", "vllm.model_executor.layers.logits_processor",
                "vllm.model_executor.layers.vocab_parallel_embedding", "vllm.model_executor.models", "vllm.model_executor.models.interfaces",
                "vllm.model_executor.models.interfaces_base", "vllm.model_executor.models.llama", "vllm.model_executor.sampling_metadata",
                "vllm.multimodal", "vllm.multimodal.inputs", "vllm.multimodal.parse", "vllm.multimodal.processing", "vllm.multimodal.profiling",
                "vllm.sequence")


def import_reference_t3():
    """models/t3/t3.py of the reference, imported in place.  vllm is not installed (ModuleNotFoundError, nothing was refused): the
    names t3.py imports from it are placeholder classes that raise on any use once the import is over (_PlaceholderMeta)."""
    import importlib
    import_reference_leaf_modules()
    assert importlib.util.find_spec("vllm") is None or isinstance(sys.modules.get("vllm"), _PlaceholderModule), "a real vllm is importable: use it instead"
    for name in VLLM_MODULES:
        m = _PlaceholderModule(name); m.__path__ = []; sys.modules[name] = m
    _Seal.sealed = False
    t3 = importlib.import_module("chatterbox_vllm.models.t3.t3")
    _Seal.sealed = True
    return t3


def _crc_rows(t: torch.Tensor) -> np.ndarray:
    """CRC-32 of every row's bytes (bf16 rows of 2048 -> 4096 bytes): a compact bit-exact pin for long prompts"""
    import zlib
    b = t.contiguous().view(torch.int16).numpy()
    return np.array([zlib.crc32(b[i].tobytes()) for i in range(b.shape[0])], dtype=np.uint32)


def g2_prompt_embeds():
    """prompt_embeds.npz -- OUTPUTS OF THE REFERENCE'S OWN CODE on the path's front end:
      * create_triangular_matrix (t3.py:94-102);
      * T3VllmModel.split_prefill_decode (t3.py:340-421) on a step's flat id vector (two prefill blocks back to back);
      * T3VllmModel.get_input_embeddings, prefill branches (t3.py:542-561 full block; :562-582 start-only chunk; :583-611 end chunk with
        the tail of the conditioning; :612-632 end chunk without conditioning, text positions from the triangular rows), bf16 modules as
        vLLM holds them (config.json torch_dtype), fp32 multimodal tensor as tts.py:286 hands it over;
      * what its decode branch (t3.py:440-486) literally returns for N = 1 and N = 2.
    The model object is made with T3VllmModel.__new__ + nn.Module.__init__ (its __init__ builds vllm's LlamaModel); the five attributes
    the methods read are set as t3.py:270-284 and :325-330 do, from the product's seeded synthetic checkpoint.  T3MultiModalProcessor.apply
    (t3.py:143-249, the id layout of :189-200) cannot run without vllm's processor base class: the id vectors below are built HERE from the
    module's own constants; only the triangular matrix comes from reference code."""
    import torch.nn as nn
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors
    t3 = import_reference_t3()
    assert not _Seal.touched
    lpe = sys.modules["chatterbox_vllm.models.t3.modules.learned_pos_emb"]
    cfg = sys.modules["chatterbox_vllm.models.t3.modules.t3_config"]
    tok = json.load(open(os.path.join(HERE, "tokenizer.json")))
    cond = synthetic_cond_emb(1)                                        # fp32 CPU tensor, as tts.py:286 ships it
    NC, OFF = t3.CONDITIONING_SIZE, t3.SPEECH_TOKEN_OFFSET
    out = {"constants": np.array([t3.PREFILL_COND_START_TOKEN, t3.PREFILL_COND_END_TOKEN, t3.PREFILL_END_TOKEN, NC, OFF], np.int32)}

    def model(vocab):
        w = {k: v for k, v in synthetic_tensors(1, vocab, 1234) if not k.startswith("tfmr.")}
        m = t3.T3VllmModel.__new__(t3.T3VllmModel)
        nn.Module.__init__(m)
        m.t3conf = cfg.T3Config(); m.dim = m.t3conf.n_channels                                  # t3.py:273-274
        m.text_emb = nn.Embedding(vocab, m.dim); m.speech_emb = nn.Embedding(m.t3conf.speech_tokens_dict_size, m.dim)   # :276-277
        m.text_pos_emb = lpe.LearnedPositionEmbeddings(m.t3conf.max_text_tokens + 2, m.dim)     # :280-281
        m.speech_pos_emb = lpe.LearnedPositionEmbeddings(m.t3conf.max_speech_tokens + 2 + 2, m.dim)   # :283-284
        m.to(torch.bfloat16)
        m.text_emb.load_state_dict({"weight": w["text_emb.weight"]}); m.speech_emb.load_state_dict({"weight": w["speech_emb.weight"]})
        m.text_pos_emb.load_state_dict({"emb.weight": w["text_pos_emb.emb.weight"]}); m.speech_pos_emb.load_state_dict({"emb.weight": w["speech_pos_emb.emb.weight"]})
        m.precomputed_text_pos_emb = m.text_pos_emb.get_fixed_embedding(torch.arange(m.t3conf.max_text_tokens + 2))[0]             # :325-326
        m.precomputed_speech_pos_emb = m.speech_pos_emb.get_fixed_embedding(torch.arange(m.t3conf.max_speech_tokens + 2 + 2))[0]   # :329-330
        return m.eval()

    def ids_and_mm(text_ids):
        ids = [t3.PREFILL_COND_START_TOKEN] + [text_ids[0]] * (NC - 2) + [t3.PREFILL_COND_END_TOKEN] + list(text_ids) + [t3.PREFILL_END_TOKEN]
        mm = torch.cat([cond, t3.create_triangular_matrix(len(text_ids), cond.shape[1]), torch.zeros(1, cond.shape[1])], dim=0)   # t3.py:212-221
        assert len(ids) == mm.shape[0]
        return torch.tensor(ids), mm

    tri = t3.create_triangular_matrix(5, 7)
    out["tri_5x7"] = tri.numpy()
    with torch.no_grad():
        # ---- full blocks of the three BASELINE prompts: per-row CRC-32 of the [T, 2048] output (cast to bf16 as vLLM's input buffer holds it)
        for key, vocab in (("en_english_ids", 704), ("en_mtl_ids", 2454), ("es_mtl_ids", 2454)):
            m = model(vocab)
            ids, mm = ids_and_mm(tok[key])
            y = m.get_input_embeddings(ids, [mm])
            assert tuple(y.shape) == (len(ids), 2048)
            out[f"full_{key}_crc"] = _crc_rows(y.to(torch.bfloat16)); out[f"full_{key}_dtype"] = np.array(str(y.dtype))
        # ---- a short prompt (11 text ids, T = 46), stored in full, and every two-chunk split of it the reference's branches cover
        m = model(2454)
        text = [635] + [int(x) for x in np.random.RandomState(2).randint(0, 2454, size=9)] + [0]
        ids, mm = ids_and_mm(text)
        T = len(ids)
        y = m.get_input_embeddings(ids, [mm])
        out["short_text_ids"] = np.array(text, np.int32); out["short_ids"] = ids.numpy().astype(np.int32)
        out["short_full"] = y.to(torch.bfloat16).view(torch.int16).numpy(); out["short_full_dtype"] = np.array(str(y.dtype))
        for k in (1, 10, 33, 34, 35, 40, T - 1):                        # chunk A = rows [0, k), chunk B = rows [k, T)
            ya = m.get_input_embeddings(ids[:k], [mm[:k]]); yb = m.get_input_embeddings(ids[k:], [mm[k:]])
            assert ya.shape[0] == k and yb.shape[0] == T - k
            out[f"short_split{k}_a_crc"] = _crc_rows(ya.to(torch.bfloat16)); out[f"short_split{k}_b_crc"] = _crc_rows(yb.to(torch.bfloat16))
        # ---- split_prefill_decode on a flat step vector: two full prefill blocks back to back (a new block starts at every 695)
        ids2, mm2 = ids_and_mm([708, 5, 6, 0])
        flat = torch.cat([ids, ids2]); parts = m.split_prefill_decode(flat, [mm, mm2])
        out["split_lengths"] = np.array([len(p[0]) for p in parts], np.int32)
        out["split_mm_rows"] = np.array([-1 if p[1] is None else p[1].shape[0] for p in parts], np.int32)
        yy = m.get_input_embeddings(flat, [mm, mm2])
        out["two_blocks_crc"] = _crc_rows(yy.to(torch.bfloat16))
        # a decode run between them (ids >= 2500, no multimodal rows): how the function segments it
        flat3 = torch.cat([ids, torch.tensor([OFF + 17]), ids2]); parts3 = m.split_prefill_decode(flat3, [mm, mm2])
        out["split3_lengths"] = np.array([len(p[0]) for p in parts3], np.int32)
        out["split3_is_decode"] = np.array([p[1] is None for p in parts3], np.int32)
        # ---- the decode branch as written: N = 1 -> a [1, 2048, 1024] tensor whose [0, j, :] is speech_emb[id] + speech_pos[j]
        # for j < 1024 (seq_len is read from the channel dimension), twice along dim 1; N = 2 -> a broadcast error
        tok_id = 4242
        d1 = m.get_input_embeddings(torch.tensor([OFF + tok_id]), None)
        out["decode_n1_shape"] = np.array(d1.shape, np.int32)
        out["decode_n1_row0"] = d1[0, 0].to(torch.bfloat16).view(torch.int16).numpy()            # = speech_emb[id] + speech_pos[0]: pos_policy = 1
        out["decode_n1_row5"] = d1[0, 5].to(torch.bfloat16).view(torch.int16).numpy()            # = speech_emb[id] + speech_pos[5]
        out["decode_n1_second_half_equal"] = np.array(bool(torch.equal(d1[0, :1024], d1[0, 1024:])))
        out["decode_token"] = np.array(tok_id, np.int32)
        try:
            m.get_input_embeddings(torch.tensor([OFF + 1, OFF + 2]), None)
            out["decode_n2_error"] = np.array("")
        except RuntimeError as ex:
            out["decode_n2_error"] = np.array(type(ex).__name__)
        # ---- decode rows of a real stream: the short prompt decoded greedily by the ORACLE (1 layer, both position policies); for the
        # token fed back at step k the reference's decode branch gives [0, 0, :] (its literal index-0 fallback = pos_policy 1) and
        # [0, k, :] (= speech_emb[tok] + speech_pos[k], the exact per-sequence position = pos_policy 0), each twice along dim 1
        from oracle import oracle as O
        om = O.OracleModel(1, 2454, max_pos=128).load(synthetic_tensors(1, 2454, 1234))
        for pol in (0, 1):
            gids, _ = om.generate(ids.tolist(), cond, O.make_sampling(temperature=0.0, repetition_penalty=1.0, max_tokens=5, ignore_eos=True, pos_policy=pol), max_model_len=128)
            out[f"short_greedy_ids_policy{pol}"] = np.array(gids, np.int32)
            rows = []
            for k in range(1, 5):                                        # decode step k embeds token k - 1 of the stream
                d = m.get_input_embeddings(torch.tensor([OFF + gids[k - 1]]), None)
                row = d[0, k if pol == 0 else 0].to(torch.bfloat16)
                rows.append(torch.cat([row, row]))
            out[f"short_decode_rows_policy{pol}_crc"] = _crc_rows(torch.stack(rows))
        om.close()
    assert not _Seal.touched, _Seal.touched                              # no executed line of any case above depended on vllm
    np.savez_compressed(os.path.join(HERE, "prompt_embeds.npz"), **out)
    print("G2 prompt_embeds:", {k: (v.shape if v.ndim else v.item()) for k, v in out.items() if "crc" not in k})

def g4_hf_gate():
    """hf_gate.npz -- the fidelity gate of tests/hf_gate.py at FULL length (128 teacher-forced decode steps, two
    prompts: 22 text ids / English vocabulary, and the 141-row es prompt of C3 / multilingual vocabulary), 30 layers, non-trivial norm
    weights, against transformers' LlamaModel in fp32 AND in bf16:
      * committed HF vectors, so that the pin does not depend on `transformers` at test time: HF-fp32 post-CFG logits [8194] and hidden
        states after 1 / 2 / 30 layers (both CFG streams) at steps 0, 1, 2 (prefill's last row, then two KV-cache decode steps);
      * the per-step evidence over all 128 steps: max / mean logit error of the oracle and of HF-bf16 against HF-fp32, greedy agreement
        and HF's top-1 / top-2 margin, overlap of the top-p = 0.8 nuclei."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hf_gate as H
    from oracle import oracle as O
    from util import make_prompt
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb
    tok = json.load(open(os.path.join(HERE, "tokenizer.json")))
    cond = synthetic_cond_emb(1)
    NL, N, SEL = 30, 128, (0, 1, 2)
    out = {"sel_steps": np.array(SEL, np.int32), "n_steps": np.int32(N)}
    for name, vocab, prompt in (("p22", 704, make_prompt(22, seed=5)), ("es", 2454, assemble_prompt_ids(tok["es_mtl_ids"]))):
        import time
        t0 = time.time()
        tens = H.gate_tensors(NL, vocab)
        ids, lg, fin, _ = H.oracle_teacher_forced(O, tens, NL, vocab, prompt, cond, N)
        m = O.OracleModel(NL, vocab, max_pos=len(prompt) + 4).load(tens)
        ec, eu = m.prompt_embeds(prompt, cond)
        gen_ids, _ = m.generate(prompt, cond, O.make_sampling(temperature=0.0, repetition_penalty=1.0, max_tokens=3, ignore_eos=True), max_model_len=len(prompt) + 4)
        m.close()
        assert gen_ids == ids[:3]                                  # the row-by-row drive is the generate loop
        t1 = time.time()
        lg32, hid32, tap32 = H.hf_teacher_forced(tens, NL, ec, eu, ids, torch.float32, taps=(1, 2))
        t2 = time.time()
        lg16, hid16, tap16 = H.hf_teacher_forced(tens, NL, ec, eu, ids, torch.bfloat16, taps=(1, 2))
        t3 = time.time()
        cmp = H.compare(lg, lg32, lg16, ids)
        # the oracle's own taps at the committed steps (for the record: error of the hidden states against HF fp32, oracle vs HF bf16)
        ids3, _, fin3, otap = H.oracle_teacher_forced(O, tens, NL, vocab, prompt, cond, len(SEL), taps=(1, 2))
        nw = dict(tens)["tfmr.norm.weight"].float()
        f32 = fin3.float(); post = f32 * torch.rsqrt(f32.pow(2).mean(-1, keepdim=True) + 1e-5) * nw
        sel = list(SEL)
        out[f"{name}_ids"] = np.array(ids, np.int16)
        out[f"{name}_prompt"] = np.array(prompt, np.int32)
        out[f"{name}_hf32_logits"] = lg32[sel].numpy()
        out[f"{name}_hf32_hidden_l1"] = tap32[1][sel].numpy(); out[f"{name}_hf32_hidden_l2"] = tap32[2][sel].numpy()
        out[f"{name}_hf32_hidden_l30_postnorm"] = hid32[sel].numpy()
        out[f"{name}_hidden_err_oracle"] = np.array([(otap[1].float() - tap32[1][sel]).abs().mean(), (otap[2].float() - tap32[2][sel]).abs().mean(), (post - hid32[sel]).abs().mean()])
        out[f"{name}_hidden_err_hfbf16"] = np.array([(tap16[1][sel] - tap32[1][sel]).abs().mean(), (tap16[2][sel] - tap32[2][sel]).abs().mean(), (hid16[sel] - hid32[sel]).abs().mean()])
        for k, v in cmp.items():
            out[f"{name}_{k}"] = np.asarray(v)
        print(f"  {name}: T = {len(prompt)}, oracle {t1 - t0:.0f} s, HF fp32 {t2 - t1:.0f} s, HF bf16 {t3 - t2:.0f} s; max logit err oracle {cmp['err_max_oracle'].max():.4f} "
              f"/ HF-bf16 {cmp['err_max_hfbf16'].max():.4f}; mean err ratio oracle / HF-bf16 max over steps {np.max(cmp['err_mean_oracle'] / cmp['err_mean_hfbf16']):.3f}; "
              f"greedy agreement {int(cmp['agree_oracle'].sum())}/{N}; nucleus Jaccard min oracle {cmp['nucleus_jaccard_oracle'].min():.4f} / HF-bf16 {cmp['nucleus_jaccard_hfbf16'].min():.4f}; "
              f"hidden err oracle {out[name + '_hidden_err_oracle']} HF-bf16 {out[name + '_hidden_err_hfbf16']}", flush=True)
    np.savez_compressed(os.path.join(HERE, "hf_gate.npz"), **out)
    print("G4 hf_gate:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


def g8_postfilter():
    """Decisions of the reference's AlignmentStreamAnalyzer (imported; run on CPU) driven by the loop of tts.py:329-350."""
    import importlib
    import_reference_leaf_modules()
    m = types.ModuleType("chatterbox_vllm.models.t3.inference"); m.__path__ = [REF_PKG + "/models/t3/inference"]
    sys.modules["chatterbox_vllm.models.t3.inference"] = m
    asa = importlib.import_module("chatterbox_vllm.models.t3.inference.alignment_stream_analyzer")
    rs = np.random.RandomState(3)
    cases = []
    def run(tokens, n_text):
        an = asa.AlignmentStreamAnalyzer(text_tokens_count=n_text, eos_token_id=6562, device="cpu")
        cleaned = []
        for tok in tokens:
            lg = an.step(torch.zeros(1, 8194), next_token=torch.tensor(tok))
            if lg[0, 6562].item() > 2 ** 14:
                break
            cleaned.append(int(tok))
        r = an.get_analysis_result()
        return cleaned, bool(r.repetition), bool(r.long_tail)
    specs = []
    for n_text in (1, 2, 4, 6, 10, 42, 60, 200):
        for length in (0, 1, 2, 3, 5, 30, 150, 400):
            toks = rs.randint(0, 6561, size=length).tolist()
            specs.append((toks, n_text))
            if length >= 5:
                t2 = list(toks); k = int(rs.randint(2, length)); t2[k] = t2[k - 1] = t2[k - 2]      # plant a triple
                specs.append((t2, n_text))
                t3 = list(toks); t3[-1] = 6562; t3[0] = 7000                                         # specials (range filter input)
                specs.append((t3, n_text))
    for toks, n_text in specs:
        cleaned, rep, tail = run(toks, n_text)
        cases.append({"tokens": toks, "text_token_count": n_text, "cleaned": cleaned, "repetition": rep, "long_tail": tail})
    json.dump(cases, open(os.path.join(HERE, "postfilter.json"), "w"))
    print("G8 postfilter:", len(cases), "cases,", sum(len(c["cleaned"]) < len(c["tokens"]) for c in cases), "truncated")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g1b", "g2", "g3", "g7", "g7b", "g6", "g8", "g9", "g6b"]
    if "g1" in which: g1_cond_enc()
    if "g1b" in which: g1b_cond_enc_synthetic()
    if "g3" in which: g3_rope()
    if "g7" in which: g7_tokenizer()
    if "g7b" in which: g7b_tokenizer_cases()
    if "g6" in which: g6_streams()
    if "g8" in which: g8_postfilter()
    if "g9" in which: g9_c4_requests()
    if "g6b" in which: g6b_streams_30_layers_multilingual()
    if "g2" in which: g2_prompt_embeds()
    if "g4" in which: g4_hf_gate()                         # not in the default list: ~10 minutes of CPU
    if "g6c" in which: g6c_full_length_streams()          # not in the default list: about an hour of CPU
