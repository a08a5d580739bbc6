"""GPU engine vs the COMMITTED golden streams (tests/golden/streams.npz, made by tests/golden/make_golden.py with the
oracle in the build container) at the BASELINE.json config shapes -- no oracle run needed on the GPU box:
  C1/C2: English vocab, T = 108 prompt, B = 1, greedy + the reference's sampling defaults (2-layer and the real 30-layer depth)
  C3:    multilingual vocab, en (T = 116) + es (T = 141) prompts in one batch
Bar: token ids equal, logits bit-identical."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    from chatterbox_vllm2_amd import engine as E
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors
    E.load_library()
    z = np.load(os.path.join(G, "streams.npz")); tok = json.load(open(os.path.join(G, "tokenizer.json")))
    return dict(E=E, z=z, tok=tok, cond=synthetic_cond_emb(1), asm=assemble_prompt_ids, syn=synthetic_tensors)


def _run(c, layers, vocab, reqs, max_model_len, want_logits_steps=()):
    E = c["E"]
    eng = E.T3Engine(n_layers=layers, text_vocab=vocab, max_model_len=max_model_len, max_seqs=max(2, len(reqs)), kv_bytes=2 << 30, debug_logits=True)
    eng.load_tensors(c["syn"](layers, vocab, 1234)); eng.finalize()
    for rid, prompt, kw in reqs:
        eng.add_request(rid, prompt, c["cond"], E.make_sampling(**kw))
    logits = {}
    step = 0
    while eng.num_unfinished():
        r = eng.step()
        if r.n_sampled and step in want_logits_steps and eng.num_unfinished():
            logits[step] = eng.debug_logits(reqs[0][0])
        step += 1 if r.n_sampled else 0
    out = {rid: [t - 2500 for t in eng.get_output(rid)[0]] for rid, _, _ in reqs}
    eng.close()
    return out, logits


def test_c1_english_two_layers(ctx):
    p = ctx["asm"](ctx["tok"]["en_english_ids"])
    assert len(p) == 108
    out, lg = _run(ctx, 2, 704, [(0, p, dict(temperature=0.0, max_tokens=64, ignore_eos=True))], 400, want_logits_steps=(0, 63))
    assert out[0] == ctx["z"]["l2_en_greedy_ids"].tolist()
    assert np.array_equal(lg[0].numpy().view(np.int32), ctx["z"]["l2_en_greedy_logits_step0"].view(np.int32))
    out, _ = _run(ctx, 2, 704, [(0, p, dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=0, max_tokens=64, ignore_eos=True))], 400)
    assert out[0] == ctx["z"]["l2_en_sampled_ids"].tolist()


def test_c3_multilingual_batch(ctx):
    en, es = ctx["asm"](ctx["tok"]["en_mtl_ids"]), ctx["asm"](ctx["tok"]["es_mtl_ids"])
    assert (len(en), len(es)) == (116, 141)
    kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, max_tokens=48, ignore_eos=True)
    out, _ = _run(ctx, 2, 2454, [(0, en, dict(uid=0, **kw)), (1, es, dict(uid=1, **kw))], 400)
    assert out[0] == ctx["z"]["l2_mtl_en_sampled_ids"].tolist()
    assert out[1] == ctx["z"]["l2_mtl_es_sampled_ids"].tolist()


def test_c2_full_depth_30_layers(ctx):
    """The real model depth (30 layers, 1.07 GB of weights): ids and first-step logits against the committed oracle stream."""
    p = ctx["asm"](ctx["tok"]["en_english_ids"])
    out, lg = _run(ctx, 30, 704, [(0, p, dict(temperature=0.0, max_tokens=16, ignore_eos=True))], 200, want_logits_steps=(0,))
    assert out[0] == ctx["z"]["l30_en_greedy_ids"].tolist()
    assert np.array_equal(lg[0].numpy().view(np.int32), ctx["z"]["l30_en_greedy_logits_step0"].view(np.int32))
    out, _ = _run(ctx, 30, 704, [(0, p, dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=0, max_tokens=16, ignore_eos=True))], 200)
    assert out[0] == ctx["z"]["l30_en_sampled_ids"].tolist()


def test_full_size_batch_invariance_c3(ctx):
    """BASELINE size (30 layers, B = 32, max_model_len 1000, graph replay), every utterance run to the END of its length (884 / 859
    tokens, contexts up to 1000): utterances 0 (en) and 17 (es) of the batch against their committed FULL-LENGTH 30-layer ORACLE
    streams (streams30_full.npz, make_golden.py g6c) -- the 64-row production schedules and the 4-wave fused attention at every
    context the bench times them at -- and an utterance's stream does not depend on its batch (B = 32 vs B = 1)."""
    E = ctx["E"]
    en, es = ctx["asm"](ctx["tok"]["en_mtl_ids"]), ctx["asm"](ctx["tok"]["es_mtl_ids"])
    eng = E.T3Engine(n_layers=30, text_vocab=2454, max_model_len=1000, max_seqs=32, kv_bytes=10 << 30, enforce_eager=False, max_batched_rows=8192)
    eng.load_tensors(ctx["syn"](30, 2454, 1234)); eng.finalize()
    kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, max_tokens=1000, ignore_eos=True)
    for i in range(32):
        eng.add_request(i, en if i < 16 else es, ctx["cond"], E.make_sampling(uid=i, **kw))
    eng.run_until_done()
    batch = {i: eng.get_output(i)[0] for i in range(32)}
    zf = np.load(os.path.join(G, "streams30_full.npz")); z30 = np.load(os.path.join(G, "streams30.npz"))
    for i in (0, 17):
        want = zf[f"c3_uid{i}_ids"].tolist()
        assert len(want) == 1000 - (116 if i < 16 else 141) and want[:24] == z30[f"c3_uid{i}_ids"].tolist()
        got = [t - 2500 for t in batch[i]]
        first = next((k for k, (a, b) in enumerate(zip(got, want)) if a != b), None)
        assert got == want, f"utterance {i} of the B=32 batch leaves the full-length 30-layer oracle stream at token {first} of {len(want)}"
    for i in range(32):
        eng.release(i)
    for i in (0, 17, 31):
        eng.add_request(100 + i, en if i < 16 else es, ctx["cond"], E.make_sampling(uid=i, **kw))
        eng.run_until_done()
        assert eng.get_output(100 + i)[0] == batch[i], f"utterance {i} differs between B=32 and B=1"
        eng.release(100 + i)
    assert len({tuple(v) for v in batch.values()}) == 32        # distinct uids -> distinct streams
    eng.close()


def test_c4_continuous_batching_full_size(ctx):
    """C4 (BASELINE.json configs[3]): the request stream tokenised from the reference's docs/benchmark-text-*.txt (c4_requests.json: 499
    sentences, en / fr / zh, G ~ U{200..800}) through 128 slots x 30 layers with graph replay and the run-ahead loop -- admission, chunked
    prefill between decode steps (the 256-row decode schedules), retirement.  Six utterances against their committed 30-layer oracle
    streams (first 32 tokens: streams are batch-invariant), every utterance runs to its length, every KV block comes back."""
    E = ctx["E"]
    c4 = json.load(open(os.path.join(G, "c4_requests.json")))
    z30 = np.load(os.path.join(G, "streams30.npz"))
    reqs = c4["requests"]
    eng = E.T3Engine(n_layers=30, text_vocab=2454, max_model_len=c4["max_model_len"], max_seqs=c4["slots"], gpu_memory_utilization=0.5,
                     enforce_eager=False, max_batched_rows=8192)
    eng.load_tensors(ctx["syn"](30, 2454, 1234)); eng.finalize()
    total_blocks = eng.stats().kv_blocks_total
    for i, r in enumerate(reqs):
        eng.add_request(i, ctx["asm"](r["text_ids"]), ctx["cond"], E.make_sampling(uid=i, max_tokens=r["max_tokens"], ignore_eos=True, **c4["sampling"]))
    eng.run_until_done()
    st = eng.stats()
    assert st.tokens_generated == sum(r["max_tokens"] for r in reqs)
    assert st.kv_blocks_free == total_blocks == st.kv_blocks_total
    zf = np.load(os.path.join(G, "streams30_full.npz"))
    checked = full = 0
    for i, r in enumerate(reqs):
        ids, fr = eng.get_output(i)
        assert len(ids) == r["max_tokens"] and fr == 2
        key = f"c4_req{i}_ids"
        if key in z30.files:
            want = z30[key].tolist()
            assert [t - 2500 for t in ids[:len(want)]] == want, f"C4 request {i} ({r['lang']}) differs from the 30-layer oracle stream"
            checked += 1
        if key in zf.files:                              # two requests (fr, 317 tokens; zh, 767 tokens) to their full max_tokens
            want = zf[key].tolist()
            assert len(want) == r["max_tokens"] and [t - 2500 for t in ids] == want, f"C4 request {i} ({r['lang']}) leaves its full-length oracle stream"
            full += 1
        eng.release(i)
    assert checked == 6 and full == 2
    eng.close()


def test_c2_b1_through_llm_as_the_server_drives_it(ctx):
    """C2 (BASELINE.json configs[1]): English vocabulary, B = 1, max_model_len = 400, 30 layers, through the vLLM-shaped surface with
    the flags the unchanged server passes (enforce_eager=True, api_server.py:157): graph replay is used regardless (llm.py), and the
    292-token utterance starts with the committed 30-layer oracle stream."""
    from chatterbox_vllm2_amd import LLM, SamplingParams
    llm = LLM(model="./t3-model", task="generate", tokenizer="EnTokenizer", tokenizer_mode="custom", gpu_memory_utilization=0.2,
              enforce_eager=True, max_model_len=400, max_num_seqs=1, load_format="dummy")
    assert llm.engine.cfg.enforce_eager == 0
    p = ctx["tok"]["en_english_ids"]
    assert len(ctx["asm"](p)) == 108
    r = llm.generate([{"prompt_token_ids": p, "multi_modal_data": {"conditionals": [ctx["cond"]]}}],
                     SamplingParams(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, max_tokens=400 - 108, ignore_eos=True))
    ids = [t - 2500 for t in r[0].outputs[0].token_ids]
    want = np.load(os.path.join(G, "streams30_full.npz"))["c2_en_sampled_ids"].tolist()
    assert len(want) == 292 and want[:16] == ctx["z"]["l30_en_sampled_ids"].tolist()
    assert ids == want and r[0].outputs[0].finish_reason == "length"
    llm.shutdown()
