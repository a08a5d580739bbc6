"""Engine-level parity through the C ABI: token ids BIT-EXACT against the oracle, post-CFG logits
bit-identical at every step (stated tolerance: 0 ulp; the contract fixes all rounding), plus the
scheduler properties the domain offers (batch invariance, continuous batching, chunked prefill)."""
import numpy as np
import pytest
import torch

from util import assert_bit_equal, make_prompt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from chatterbox_vllm2_amd import engine
    engine.load_library()
    return engine


@pytest.fixture(scope="module")
def cond():
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb
    return synthetic_cond_emb(1)


@pytest.fixture(scope="module")
def tiny_engine(E, tiny_weights):
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=400, max_seqs=8, kv_bytes=1 << 30, debug_logits=True)
    eng.load_tensors(tiny_weights); eng.finalize()
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def tiny_oracle(oracle, tiny_weights):
    m = oracle.OracleModel(2, 704, max_pos=400, n_streams=2).load(tiny_weights)
    yield m
    m.close()


CASES = [dict(temperature=0.0), dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=5),
         dict(temperature=0.8, top_p=1.0, repetition_penalty=2.0, seed=6, pos_policy=1)]


@pytest.mark.parametrize("kw", CASES)
def test_single_utterance_ids_and_logits(E, oracle, tiny_engine, tiny_oracle, cond, kw):
    prompt = make_prompt(20, seed=1)
    n = 40
    want, want_lg = tiny_oracle.generate(prompt, cond, oracle.make_sampling(max_tokens=n, ignore_eos=True, uid=3, **kw), want_logits=True, max_model_len=400)
    tiny_engine.add_request(100, prompt, cond, E.make_sampling(max_tokens=n, ignore_eos=True, uid=3, **kw))
    step = 0
    while tiny_engine.num_unfinished():
        r = tiny_engine.step()
        if r.n_sampled and tiny_engine.num_unfinished():
            assert_bit_equal(tiny_engine.debug_logits(100), want_lg[step], f"logits step {step}")
        step += r.n_sampled
    got, fr = tiny_engine.get_output(100)
    tiny_engine.release(100)
    assert fr == 2
    assert [t - 2500 for t in got] == want


def test_stop_token_and_limits(E, oracle, tiny_engine, tiny_oracle, cond):
    prompt = make_prompt(12, seed=2)
    ref, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(temperature=0.0, max_tokens=30, ignore_eos=True), max_model_len=400)
    stop = ref[7]                     # pretend the 8th greedy token is the stop id
    first = ref.index(stop)
    want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(temperature=0.0, max_tokens=30, stop_token=stop), max_model_len=400)
    assert want == ref[: first + 1]
    tiny_engine.add_request(1, prompt, cond, E.make_sampling(temperature=0.0, max_tokens=30, stop_token=stop))
    tiny_engine.run_until_done()
    got, fr = tiny_engine.get_output(1); tiny_engine.release(1)
    assert fr == 1 and [t - 2500 for t in got] == want          # the stop id is emitted


def test_batch_invariance_and_continuous_batching(E, oracle, tiny_engine, tiny_oracle, cond):
    """10 utterances of different prompt lengths and lengths through 8 slots: admit/retire between steps;
    every stream must equal its single-utterance oracle stream (independence of batch composition)."""
    rs = np.random.RandomState(0)
    reqs = []
    for i in range(10):
        prompt = make_prompt(int(rs.randint(3, 60)), seed=10 + i)
        n = int(rs.randint(5, 45))
        kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=99, uid=i, max_tokens=n, ignore_eos=True)
        reqs.append((i, prompt, kw))
        tiny_engine.add_request(i, prompt, cond, E.make_sampling(**kw))
    tiny_engine.run_until_done()
    for i, prompt, kw in reqs:
        got, _ = tiny_engine.get_output(i); tiny_engine.release(i)
        want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=400)
        assert [t - 2500 for t in got] == want, f"utterance {i}"
    st = tiny_engine.stats()
    assert st.kv_blocks_free == st.kv_blocks_total      # every block returned


@pytest.mark.parametrize("n_groups,eager", [(1, True), (2, False), (3, False), (4, True)])        # (1, False) is every other test's configuration
def test_groups_and_graph_replay_do_not_change_ids(E, oracle, tiny_weights, tiny_oracle, cond, n_groups, eager):
    """Concurrent utterance groups (separate streams) and hipGraph replay are scheduling only: every stream still
    equals its single-utterance oracle stream.  Requests finish at different steps, so graphs are re-captured."""
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=400, max_seqs=6, kv_bytes=1 << 29, n_groups=n_groups, enforce_eager=eager)
    eng.load_tensors(tiny_weights); eng.finalize()
    reqs = []
    for i in range(7):
        prompt = make_prompt(5 + 7 * i, seed=40 + i)
        kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=3, uid=i, max_tokens=6 + 5 * (i % 4), ignore_eos=True)
        reqs.append((i, prompt, kw)); eng.add_request(i, prompt, cond, E.make_sampling(**kw))
    eng.run_until_done()
    for i, prompt, kw in reqs:
        got, _ = eng.get_output(i)
        want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=400)
        assert [t - 2500 for t in got] == want, f"utterance {i} (groups={n_groups}, eager={eager})"
    eng.close()


@pytest.mark.parametrize("switches", [{"T3_ZERO_COPY": "0"}, {"T3_PREFETCH": "0", "T3_GEMM_SMALL_M": "0"}, {"T3_PREFETCH_DOWN_LINES": "1024"}, {"T3_LONGEST_FIRST": "0"}])
def test_transport_and_prefetch_switches_do_not_change_ids(E, oracle, tiny_weights, tiny_oracle, cond, switches, monkeypatch):
    """How a step's metadata and ids travel (read / written in pinned host memory by the step's own kernels, or by copy kernels), whether
    gate/up's epilogue waves fetch down_proj's weights into L2, and whether the few-row GEMM forms skip padded activation rows are
    transport and scheduling only (as is the order of a step's decode rows: longest context first by default, admission order with
    T3_LONGEST_FIRST=0): with each of them switched off (the defaults are on and run in every other test) every stream --
    1, 2 and 5 utterances at a time, graph replay with run-ahead, utterances that stop at different steps -- still equals its oracle stream."""
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=400, max_seqs=5, kv_bytes=1 << 29, enforce_eager=False)
    eng.load_tensors(tiny_weights); eng.finalize()
    rid = 0
    for wave in (1, 2, 5):
        reqs = []
        for i in range(wave):
            prompt = make_prompt(6 + 9 * i, seed=90 + rid)
            kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=5, uid=rid, max_tokens=9 + 4 * i, ignore_eos=True)
            reqs.append((rid, prompt, kw)); eng.add_request(rid, prompt, cond, E.make_sampling(**kw)); rid += 1
        eng.run_until_done()
        for r, prompt, kw in reqs:
            got, _ = eng.get_output(r)
            want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=400)
            assert [t - 2500 for t in got] == want, f"utterance {r} with {switches}"
    eng.close()


def test_profile_mode_times_every_class_and_changes_no_id(E, oracle, tiny_weights, tiny_oracle, cond):
    """bench.py's roofline pass: in profile mode every decode-path launch carries two HIP events as its own start / stop events
    (hipExtLaunchKernelGGL).  The ids must not change, every class must report one launch per layer and decode step (one per step for the
    head, the sampler and the embed kernel), and a launch's duration must be a kernel's duration: above zero, far below a millisecond;
    with events on ONE class only, the others report nothing."""
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=400, max_seqs=3, kv_bytes=1 << 29, enforce_eager=False)
    eng.load_tensors(tiny_weights); eng.finalize()
    reqs = []
    for i in range(3):
        prompt = make_prompt(6 + 5 * i, seed=120 + i)
        kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=9, uid=i, max_tokens=12, ignore_eos=True)
        reqs.append((i, prompt, kw)); eng.add_request(i, prompt, cond, E.make_sampling(**kw))
    while eng.stats().decode_steps == 0 or eng.stats().prefill_rows < sum(2 * len(p) for _, p, _ in reqs):
        eng.step()                                      # every prompt prefilled: what follows are decode-only steps
    eng.reset_stats(); eng.set_profile(True)
    eng.run_steps(5)
    steps = eng.stats().decode_steps
    assert steps == 5
    per_layer = {"gemm_qkv", "gemm_o", "gemm_gateup", "gemm_down", "attention"}
    for k in E.KERNEL_CLASSES:
        ms, n = eng.kernel_ms(k)
        want = 0 if k == "rope_kv" else (2 * steps if k in per_layer else steps)
        assert n == want, f"{k}: {n} launches timed, expected {want}"
        if n:
            assert 0.0005 < ms < 0.5, f"{k}: {ms} ms per launch"
    eng.reset_stats(); eng.set_profile(True, only="attention")
    eng.run_steps(3)
    for k in E.KERNEL_CLASSES:
        ms, n = eng.kernel_ms(k)
        assert (n == 6 and 0.0005 < ms < 0.5) if k == "attention" else n == 0, f"{k}: {n} launches, {ms} ms with events on the attention only"
    eng.set_profile(False)
    eng.run_until_done()
    for i, prompt, kw in reqs:
        got, _ = eng.get_output(i)
        want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=400)
        assert [t - 2500 for t in got] == want, f"utterance {i}"
    eng.close()


@pytest.mark.parametrize("run_ahead,burst", [("1", "4"), ("0", "4"), ("1", "1"), ("1", "3")])
def test_run_ahead_with_stop_tokens(E, oracle, tiny_weights, tiny_oracle, cond, run_ahead, burst, monkeypatch):
    """The C++ step loop schedules the next item -- a step, or a burst of up to 4 decode steps replayed as ONE graph (T3_STEPS_PER_GRAPH) --
    before it has read the current one's tokens (DESIGN.md "Run-ahead"): an utterance that emits its stop id has discarded row pairs in
    flight (one with single steps, up to 2 * burst - 1 with bursts) and its slot is freed when the last of them is back.  Streams must
    still end exactly at the stop id, later admissions must reuse the slots, and every KV block must come back."""
    monkeypatch.setenv("T3_RUN_AHEAD", run_ahead)
    monkeypatch.setenv("T3_STEPS_PER_GRAPH", burst)
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=400, max_seqs=4, kv_bytes=1 << 29, enforce_eager=False)
    eng.load_tensors(tiny_weights); eng.finalize()
    reqs = []
    for i in range(6):
        prompt = make_prompt(4 + 5 * i, seed=70 + i)
        kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=11, uid=i, max_tokens=20)
        ref, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(ignore_eos=True, **kw), max_model_len=400)
        stop = ref[3 + 3 * i] if i < 5 else 8193       # stops at different steps; the last one runs into max_tokens
        want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(stop_token=stop, **kw), max_model_len=400)
        assert want == ref[: ref.index(stop) + 1] if stop in ref else want == ref
        reqs.append((i, want, 1 if stop in ref else 2))
        eng.add_request(i, prompt, cond, E.make_sampling(stop_token=stop, **kw))
    done = 0
    while eng.num_unfinished():                        # bounded calls: the loop must drain its in-flight step at every return
        n = eng.run_steps(7)
        assert n > 0
        done += n
    for i, want, reason in reqs:
        got, fr = eng.get_output(i)
        assert [t - 2500 for t in got] == want and fr == reason, f"utterance {i} (run_ahead={run_ahead}, burst={burst})"
    st = eng.stats()
    assert st.kv_blocks_free == st.kv_blocks_total
    eng.close()


@pytest.mark.parametrize("burst", ["4", "2"])
def test_bursts_run_exactly_the_steps_asked_for(E, oracle, tiny_weights, tiny_oracle, cond, burst, monkeypatch):
    """t3_run_steps(n) with several decode steps per graph replay: exactly n steps whatever n mod burst is (bench.py times EXACTLY K
    steps), one token per utterance and step, length limits met in the middle of a burst window, ids equal to the oracle's."""
    monkeypatch.setenv("T3_STEPS_PER_GRAPH", burst)
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=400, max_seqs=3, kv_bytes=1 << 29, enforce_eager=False)
    eng.load_tensors(tiny_weights); eng.finalize()
    reqs = []
    for i in range(3):
        prompt = make_prompt(5 + 3 * i, seed=20 + i)
        kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=2, uid=i, max_tokens=23 + 2 * i, ignore_eos=True)      # limits 23, 25, 27
        reqs.append((i, prompt, kw)); eng.add_request(i, prompt, cond, E.make_sampling(**kw))
    assert eng.run_steps(1) == 1                       # the prefill step samples token 0 of all three
    eng.reset_stats()
    for n in (1, 5, 7, 2, 6):                          # 21 decode steps: 22 tokens each, nobody at its limit yet
        assert eng.run_steps(n) == n
    st = eng.stats()
    assert st.steps == 21 and st.decode_steps == 21 and st.tokens_generated == 63
    ms, rows = eng.step_times(64)
    assert len(ms) == 21 and (rows == 6).all() and (ms > 0).all()
    eng.run_until_done()
    for i, prompt, kw in reqs:
        got, fr = eng.get_output(i)
        want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=400)
        assert [t - 2500 for t in got] == want and fr == 2, f"utterance {i} (burst={burst})"
    st = eng.stats()
    assert st.kv_blocks_free == st.kv_blocks_total
    eng.close()


def test_decode_across_kv_block_boundaries(E, oracle, tiny_weights, cond):
    """Physical KV blocks hold 256 tokens: decode steps whose newest position crosses 255 -> 256 and 511 -> 512 (fused RoPE /
    KV-write / attention path, block-table lookups, last-tile patching) against the oracle, ragged batch, sampled and greedy."""
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=560, max_seqs=4, kv_bytes=1 << 30, enforce_eager=False)
    eng.load_tensors(tiny_weights); eng.finalize()
    m = oracle.OracleModel(2, 704, max_pos=560, n_streams=2).load(tiny_weights)
    reqs = []
    for i, (n_text, kw) in enumerate([(211, dict(temperature=0.0)), (466, dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=2, uid=1)),
                                      (150, dict(temperature=0.8, top_p=0.9, repetition_penalty=1.3, seed=2, uid=2))]):
        prompt = make_prompt(n_text, seed=90 + i)
        T = len(prompt)
        n = {0: 24, 1: 24, 2: 100}[i]                 # 246..270, 501..525, 185..285
        assert T + n < 560
        kw = dict(max_tokens=n, ignore_eos=True, **kw)
        reqs.append((i, prompt, kw))
        eng.add_request(i, prompt, cond, E.make_sampling(**kw))
    assert len(reqs[0][1]) < 256 < len(reqs[0][1]) + 24 and len(reqs[1][1]) < 512 < len(reqs[1][1]) + 24 and len(reqs[2][1]) < 256 < len(reqs[2][1]) + 100
    eng.run_until_done()
    for i, prompt, kw in reqs:
        got, _ = eng.get_output(i)
        want, _ = m.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=560)
        assert [t - 2500 for t in got] == want, f"utterance {i}"
    m.close(); eng.close()


def test_long_prompt_many_blocks(E, oracle, tiny_weights, cond):
    """A 1100-token prompt (1065 text ids: text positions up to 1064, five 256-token KV blocks per stream) prefilled in
    chunks of at most 512 rows, then decoded: ids against the oracle."""
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=1200, max_seqs=2, kv_bytes=1 << 30, max_batched_rows=1024, enforce_eager=False)
    eng.load_tensors(tiny_weights); eng.finalize()
    prompt = make_prompt(1065, seed=5)
    assert len(prompt) == 1100
    kw = dict(temperature=0.0, max_tokens=6, ignore_eos=True)
    eng.add_request(0, prompt, cond, E.make_sampling(**kw))
    eng.run_until_done()
    got, _ = eng.get_output(0)
    m = oracle.OracleModel(2, 704, max_pos=1200, n_streams=2).load(tiny_weights)
    want, _ = m.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=1200)
    assert [t - 2500 for t in got] == want
    with pytest.raises(ValueError):                    # longer than max_model_len (and than the 2050 learned text positions)
        eng.add_request(1, make_prompt(2052, seed=6), cond, E.make_sampling(max_tokens=1))
    m.close(); eng.close()


def test_many_batch_shapes_graph_cache_turnover(E, oracle, tiny_weights, tiny_oracle, cond):
    """150 short utterances through 72 slots: the number of running utterances sweeps 72 -> 0, so more than 64 distinct
    decode-step shapes are captured and the graph cache is turned over while the run-ahead step is in flight."""
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=120, max_seqs=72, kv_bytes=1 << 30, enforce_eager=False)
    eng.load_tensors(tiny_weights); eng.finalize()
    rs = np.random.RandomState(5)
    reqs = []
    for i in range(150):
        prompt = make_prompt(int(rs.randint(3, 14)), seed=300 + i)
        kw = dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=4, uid=i, max_tokens=int(rs.randint(2, 60)), ignore_eos=True)
        reqs.append((i, prompt, kw)); eng.add_request(i, prompt, cond, E.make_sampling(**kw))
    eng.run_until_done()
    for i, prompt, kw in reqs[::21]:
        want, _ = tiny_oracle.generate(prompt, cond, oracle.make_sampling(**kw), max_model_len=120)
        assert [t - 2500 for t in eng.get_output(i)[0]] == want, f"utterance {i}"
    assert all(len(eng.get_output(i)[0]) == kw["max_tokens"] for i, _, kw in reqs)
    st = eng.stats()
    assert st.kv_blocks_free == st.kv_blocks_total
    eng.close()


def test_chunked_prefill_equals_whole(E, tiny_weights, cond):
    """A row budget smaller than one prompt forces the prompt through several steps; ids must not change."""
    outs = []
    for budget in (0, 64):
        eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=300, max_seqs=2, kv_bytes=1 << 28, max_batched_rows=budget)
        eng.load_tensors(tiny_weights); eng.finalize()
        eng.add_request(0, make_prompt(100, seed=4), cond, E.make_sampling(temperature=0.0, max_tokens=12, ignore_eos=True))
        eng.run_until_done()
        outs.append(eng.get_output(0)[0]); eng.close()
    assert outs[0] == outs[1]


def test_error_behaviour(E, tiny_engine, cond):
    sp = E.make_sampling(max_tokens=4)
    with pytest.raises(ValueError):
        tiny_engine.add_request(7, [1, 2, 3], cond, sp)                       # not the 695..696..697 layout
    with pytest.raises(ValueError):
        tiny_engine.add_request(7, make_prompt(390, seed=1), cond, sp)        # does not fit max_model_len
    bad = make_prompt(10, seed=1); bad[40] = 5000
    with pytest.raises(ValueError):
        tiny_engine.add_request(7, bad, cond, sp)                             # text id out of range
    with pytest.raises(ValueError):
        tiny_engine.add_request(7, make_prompt(10, seed=1), cond[:10], sp)    # wrong conditioning shape
    assert tiny_engine.num_unfinished() == 0


def _toy_tokenizer_file(path):
    """A small character-level `tokenizers` file with the special tokens the English path relies on (ids 255 / 0 / 2 as in the
    reference's vocabulary); the reference's own tokenizer.json does not travel to the GPU box."""
    from tokenizers import Regex, Tokenizer, models, pre_tokenizers
    vocab = {f"[FILL{i}]": i for i in range(256)}
    vocab.pop("[FILL0]"); vocab.pop("[FILL1]"); vocab.pop("[FILL2]"); vocab.pop("[FILL255]")
    vocab.update({"[STOP]": 0, "[UNK]": 1, "[SPACE]": 2, "[START]": 255})
    for k, ch in enumerate("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ.,!?'-"):
        vocab.pop(f"[FILL{10 + k}]"); vocab[ch] = 10 + k
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.add_special_tokens(["[START]", "[STOP]", "[SPACE]", "[UNK]"])
    tok.pre_tokenizer = pre_tokenizers.Split(Regex("."), "isolated")
    tok.save(str(path))
    return str(path)


def test_llm_surface(E, oracle, tiny_oracle, cond, tmp_path):
    """The vLLM-shaped API used at tts.py:445-492, checked against the oracle and the committed goldens."""
    import json, os
    from chatterbox_vllm2_amd import LLM, SamplingParams
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(G, "streams.npz")); tokj = json.load(open(os.path.join(G, "tokenizer.json")))
    llm = LLM(model="./t3-model", task="generate", tokenizer="EnTokenizer", tokenizer_mode="custom", gpu_memory_utilization=0.2,
              enforce_eager=True, max_model_len=240, max_num_seqs=4, load_format="dummy", num_hidden_layers=2,
              tokenizer_file=_toy_tokenizer_file(tmp_path / "toy_tokenizer.json"))
    mm = {"conditionals": [cond]}
    # (1) committed goldens (C1 prompt, 2 layers): greedy, and the reference's sampling defaults with an explicit seed
    c1 = {"prompt_token_ids": tokj["en_english_ids"], "multi_modal_data": mm}
    r = llm.generate([c1], SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=64, ignore_eos=True))     # the golden's penalty (tts.py:416)
    assert [t - 2500 for t in r[0].outputs[0].token_ids] == z["l2_en_greedy_ids"].tolist()
    sp_seeded = SamplingParams(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, max_tokens=64, ignore_eos=True)
    r = llm.generate([c1], sp_seeded)
    assert [t - 2500 for t in r[0].outputs[0].token_ids] == z["l2_en_sampled_ids"].tolist()
    assert llm.generate([c1, c1], sp_seeded)[1].outputs[0].token_ids == r[0].outputs[0].token_ids      # per-request seed: position-independent
    # (2) a prompt STRING through the tokenizer front-end, against the oracle on the same ids (tts.py:435: "[START]" + text + "[STOP]")
    text = "[START]Hello there, world![STOP]"
    ids = llm.get_tokenizer().encode(text)
    assert ids[0] == 255 and ids[-1] == 0 and ids.count(2) == 2
    sp = SamplingParams(temperature=0.8, stop_token_ids=[6562 + 2500], max_tokens=min(1000, 240), top_p=0.8, repetition_penalty=2.0)   # tts.py:455-464
    u0 = llm._next_uid                                        # unseeded requests take consecutive RNG streams
    res = llm.generate([{"prompt": text, "multi_modal_data": mm}] * 2, sampling_params=sp)
    assert len(res) == 2
    for k, r in enumerate(res):
        want, _ = tiny_oracle.generate(assemble_prompt_ids(ids), cond, oracle.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=u0 + k,
                                                                                          max_tokens=240), max_model_len=240)
        o = r.outputs[0]
        assert [t - 2500 for t in o.token_ids] == want and min(o.token_ids) >= 2500
        assert o.finish_reason == ("stop" if want[-1] == 6562 else "length")
    # unseeded requests take fresh RNG streams: the same text again is a different utterance (the reference sets no seed)
    again = llm.generate([{"prompt": text, "multi_modal_data": mm}], sampling_params=sp)
    assert all(again[0].outputs[0].token_ids != r.outputs[0].token_ids for r in res)
    # (3) a bad prompt in the middle of a list: ValueError (HTTP 400, api_server.py:323-326) and NOTHING stays queued
    good = {"prompt": text, "multi_modal_data": mm}
    for bad in ({"prompt_token_ids": [255, 9999, 0], "multi_modal_data": mm}, {"prompt": text, "multi_modal_data": {"conditionals": [cond[:10]]}}, "plain string"):
        with pytest.raises(ValueError):
            llm.generate([good, good, bad, good], sp)
        assert llm.engine.num_unfinished() == 0
    st = llm.engine.stats()
    assert st.kv_blocks_free == st.kv_blocks_total
    nxt = llm.generate([good], SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True))       # the engine is still usable and empty
    assert len(nxt[0].outputs[0].token_ids) == 8
    assert llm.engine.cfg.enforce_eager == 0                  # enforce_eager=True is accepted and not applied by default ...
    llm.shutdown()
    eager = LLM(model="./t3-model", tokenizer="EnTokenizer", enforce_eager=True, honor_enforce_eager=True, max_model_len=240, max_num_seqs=4,
                load_format="dummy", num_hidden_layers=2, kv_cache_bytes=1 << 28)
    assert eager.engine.cfg.enforce_eager == 1                # ... unless the caller asks for vLLM's meaning: launch by launch, same ids
    r2 = eager.generate([c1], SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=64, ignore_eos=True))
    assert [t - 2500 for t in r2[0].outputs[0].token_ids] == z["l2_en_greedy_ids"].tolist()
    eager.shutdown()


def test_abort_request_states(E, tiny_engine, cond):
    """t3_abort_request: a waiting request leaves the queue, a running one frees its slot and blocks, unknown ids are reported."""
    sp = E.make_sampling(temperature=0.0, max_tokens=50, ignore_eos=True)
    free0 = tiny_engine.stats().kv_blocks_free
    for rid in (900, 901, 902):
        tiny_engine.add_request(rid, make_prompt(10, seed=rid), cond, sp)
    tiny_engine.abort(901)                                   # still waiting
    tiny_engine.step(); tiny_engine.step()                   # 900 and 902 are decoding now
    assert tiny_engine.num_unfinished() == 2 and tiny_engine.stats().kv_blocks_free < free0
    tiny_engine.abort(900)
    assert tiny_engine.num_unfinished() == 1
    tiny_engine.run_until_done()
    assert len(tiny_engine.get_output(902)[0]) == 50
    tiny_engine.release(902)
    assert tiny_engine.stats().kv_blocks_free == free0
    with pytest.raises(E.T3Error):
        tiny_engine.abort(900)                               # already gone


def test_non_finite_conditioning_is_rejected(E, tiny_engine, cond):
    bad = cond.clone(); bad[3, 7] = float("nan")
    with pytest.raises(ValueError):
        tiny_engine.add_request(950, make_prompt(10, seed=1), bad, E.make_sampling())
    bad[3, 7] = float("inf")
    with pytest.raises(ValueError):
        tiny_engine.add_request(950, make_prompt(10, seed=1), bad, E.make_sampling())
    assert tiny_engine.num_unfinished() == 0


def test_real_checkpoint_path_safetensors(E, cond, tmp_path):
    """`LLM(model=<dir>)` -> weights.iter_safetensors -> t3_load_tensor name routing (t3.py:300-332), with the extra tensors a real
    checkpoint carries (cond_enc.*, text_head.*, tfmr.embed_tokens.*) skipped like t3.py:316-319; ids equal the committed goldens."""
    import json, os
    from safetensors.torch import save_file
    from chatterbox_vllm2_amd import LLM, SamplingParams
    from chatterbox_vllm2_amd.weights import synthetic_cond_enc_tensors, synthetic_tensors
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(G, "streams.npz")); tokj = json.load(open(os.path.join(G, "tokenizer.json")))
    sd = {k: v.contiguous() for k, v in synthetic_tensors(2, 704, 1234)}
    sd.update({k: v for k, v in synthetic_cond_enc_tensors(4321)})
    sd["text_head.weight"] = torch.zeros(704, 1024, dtype=torch.bfloat16)
    sd["tfmr.embed_tokens.weight"] = torch.zeros(8, 1024, dtype=torch.bfloat16)
    mdir = tmp_path / "t3-model"; mdir.mkdir()
    save_file(sd, str(mdir / "model.safetensors"))
    llm = LLM(model=str(mdir), task="generate", tokenizer="EnTokenizer", tokenizer_mode="custom", gpu_memory_utilization=0.2,
              enforce_eager=True, max_model_len=400, max_num_seqs=2, num_hidden_layers=2)
    r = llm.generate([{"prompt_token_ids": tokj["en_english_ids"], "multi_modal_data": {"conditionals": [cond]}}],
                     SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=64, ignore_eos=True))
    assert [t - 2500 for t in r[0].outputs[0].token_ids] == z["l2_en_greedy_ids"].tolist()
    llm.shutdown()
    with pytest.raises(ValueError):
        LLM(model=str(tmp_path / "nowhere"), tokenizer="EnTokenizer", num_hidden_layers=2)        # no checkpoint, no load_format="dummy"


_DP_GPU_WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
from chatterbox_vllm2_amd import LLM, SamplingParams
from chatterbox_vllm2_amd.dp import generate_data_parallel
from chatterbox_vllm2_amd.weights import synthetic_cond_emb
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from util import make_prompt
cond = synthetic_cond_emb(1)
llm = LLM(model="", tokenizer="EnTokenizer", load_format="dummy", num_hidden_layers=2, max_model_len=200, max_num_seqs=4, kv_cache_bytes=1 << 28,
          enforce_eager=False, device_id=0)
prompts = [{"prompt_token_ids": make_prompt(6 + 3 * i, seed=40 + i)[34:-1], "multi_modal_data": {"conditionals": [cond]}} for i in range(7)]
sps = [SamplingParams(temperature=0.8, top_p=0.8, repetition_penalty=2.0, max_tokens=12 + 5 * (i % 3), ignore_eos=True, seed=(77 if i == 3 else None)) for i in range(7)]
res = generate_data_parallel(llm, prompts, sps, rank, world)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps(res))
print("SEEDED " + json.dumps(llm.generate([prompts[3]], [sps[3]])[0].outputs[0].token_ids))      # the seeded request on its own
llm.shutdown()
'''


def test_data_parallel_llm_world2_equals_world1(tmp_path):
    """the utterance list sharded over two engine processes (gloo between them, both on this box's one GPU) emits
    exactly the ids of the one-process run -- RNG streams are keyed by the global utterance index, nothing crosses ranks in a step."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dpw.py"; script.write_text(_DP_GPU_WORKER)
    def run(world, port):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        seeded = [json.loads([l for l in o.splitlines() if l.startswith("SEEDED ")][-1][7:]) for o in outs]
        res = [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]) for o in outs]
        for r, sd in zip(res, seeded):
            assert r[3] == sd, "a seeded request must give the same ids through generate_data_parallel and through LLM.generate"
        return res
    one = run(1, 29571)[0]
    two = run(2, 29572)
    assert two[0] == one and two[1] == one
    assert len(one) == 7 and all(len(t) == 12 + 5 * (i % 3) for i, t in enumerate(one))


def test_handoff_keeps_ids_on_the_device(E, cond):
    """f4 (tts.py:483-514 as one call): 9 utterances through 3 slots (slots are reused while earlier utterances wait for the hand-off);
    the padded device batch equals the host post-filter (pinned to the reference's analyzer) applied to the ids `LLM.generate` returned."""
    from chatterbox_vllm2_amd import LLM, SamplingParams
    from chatterbox_vllm2_amd.postfilter import analyze_and_clean_tokens
    llm = LLM(model="", tokenizer="EnTokenizer", load_format="dummy", num_hidden_layers=2, max_model_len=300, max_num_seqs=3, kv_cache_bytes=1 << 28)
    prompts = [{"prompt_token_ids": make_prompt(6 + 2 * i, seed=70 + i)[34:-1], "multi_modal_data": {"conditionals": [cond]}} for i in range(9)]
    sps = [SamplingParams(temperature=0.8 if i % 3 else 0.0, top_p=0.8, repetition_penalty=2.0 if i % 2 else 1.0, max_tokens=40 + 23 * i, stop_token_ids=[9062]) for i in range(9)]
    outs = llm.generate(prompts, sps, keep_for_handoff=True)
    counts = [2 * (3 + i) for i in range(9)]                      # tts.py:496: len(prompt.split()) * 2
    toks, lens = llm.handoff_tokens(outs, counts)
    assert toks.is_cuda and toks.dtype == torch.int32 and tuple(toks.shape) == (9, max(len(o.outputs[0].token_ids) for o in outs))
    toks, lens = toks.cpu(), lens.cpu()
    for i, o in enumerate(outs):
        want, _ = analyze_and_clean_tokens([t - 2500 for t in o.outputs[0].token_ids], counts[i], range_filter=True)
        assert toks[i, : int(lens[i])].tolist() == want and not toks[i, int(lens[i]):].any()
    assert any(int(lens[i]) < len(o.outputs[0].token_ids) for i, o in enumerate(outs))        # greedy streams repeat: the filter did cut something
    assert llm.engine.num_unfinished() == 0
    with pytest.raises(E.T3Error):
        llm.engine.handoff_tokens([int(outs[0].request_id)], [4])                              # released by the hand-off
    llm.shutdown()


def test_pop_finished_enumerates_more_than_a_step_result_holds(E, tiny_weights, cond):
    """T3StepResult carries the first 64 finished ids of a step; a 128-utterance step can retire more: t3_pop_finished hands out all of
    them, in order, once."""
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=64, max_seqs=80, kv_bytes=1 << 30, max_batched_rows=8192)
    eng.load_tensors(tiny_weights); eng.finalize()
    for i in range(80):
        eng.add_request(1000 + i, make_prompt(6, seed=i), cond, E.make_sampling(temperature=0.0, max_tokens=1, ignore_eos=True))
    r = eng.step()                                               # every prompt prefills in this step and samples its only token
    assert r.n_finished == 80 and r.n_sampled == 80 and list(r.finished_ids[:64]) == [1000 + i for i in range(64)]
    assert eng.pop_finished(50) == [1000 + i for i in range(50)]
    assert eng.pop_finished() == [1000 + i for i in range(50, 80)]
    assert eng.pop_finished() == []
    for i in range(80):
        assert len(eng.get_output(1000 + i)[0]) == 1
        eng.release(1000 + i)
    eng.close()


def test_handoff_needs_reservation_and_says_so(E, tiny_weights, cond):
    """Device-resident ids are kept only after t3_reserve_handoff (the default path pays nothing for the hand-off); asking for a
    hand-off without it names the cause, and reserving fewer buffers than utterances finish still works (the pool grows)."""
    eng = E.T3Engine(n_layers=2, text_vocab=704, max_model_len=128, max_seqs=4, kv_bytes=1 << 28)
    eng.load_tensors(tiny_weights); eng.finalize()
    sp = E.make_sampling(temperature=0.0, max_tokens=6, ignore_eos=True)
    eng.add_request(0, make_prompt(6, seed=1), cond, sp); eng.run_until_done()
    with pytest.raises(E.T3Error, match="reserve_handoff"):
        eng.handoff_tokens([0], [4])
    eng.release(0)
    eng.reserve_handoff(1)
    for i in range(1, 6):
        eng.add_request(i, make_prompt(6 + i, seed=i), cond, sp)
    eng.run_until_done()
    toks, lens = eng.handoff_tokens(list(range(1, 6)), [40] * 5, range_filter=False)
    for i in range(1, 6):
        assert toks[i - 1, : int(lens[i - 1])].tolist() == [t - 2500 for t in eng.get_output(i)[0]][: int(lens[i - 1])] and int(lens[i - 1]) > 0
        eng.release(i)
    eng.close()
