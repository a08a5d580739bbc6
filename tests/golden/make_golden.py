"""Generates the committed golden fixtures under tests/golden/ (run in the BUILD container, where
/root/reference is mounted; the GPU box only sees the resulting .npz / .json files).

    python tests/golden/make_golden.py

What comes from where:
  G1  cond_enc.npz     -- output of the REFERENCE's own T3CondEnc (+Perceiver, LearnedPositionEmbeddings),
                          imported from /root/reference with stub parent packages (the package
                          __init__ files import vllm, which is not installed), seeded weights + inputs.
  G3  rope.npz         -- inv_freq and cos/sin of transformers' LlamaRotaryEmbedding for the reference's
                          t3-model/config.json rope settings.
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
    which = sys.argv[1:] or ["g1", "g1b", "g3", "g7", "g7b", "g6", "g8", "g9", "g6b"]
    if "g1" in which: g1_cond_enc()
    if "g1b" in which: g1b_cond_enc_synthetic()
    if "g3" in which: g3_rope()
    if "g7" in which: g7_tokenizer()
    if "g7b" in which: g7b_tokenizer_cases()
    if "g6" in which: g6_streams()
    if "g8" in which: g8_postfilter()
    if "g9" in which: g9_c4_requests()
    if "g6b" in which: g6b_streams_30_layers_multilingual()
    if "g6c" in which: g6c_full_length_streams()          # not in the default list: about an hour of CPU
