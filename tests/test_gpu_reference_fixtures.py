"""The HIP path against OUTPUTS OF THE REFERENCE'S OWN CODE (tests/golden/prompt_embeds.npz, made by make_golden.py g2 from
models/t3/t3.py imported in the build container): the rows embed_kernel hands to the Llama blocks, as the scheduler's row descriptors
make it build them, must be bit for bit what T3VllmModel.get_input_embeddings returns (t3.py:424-647) -- the full prefill block
(:542-561), every two-chunk split its chunked branches cover (:562-632) and the decode rows (:440-486, both readings of SURVEY.md 9 Q1).
Read back through the C ABI (t3_debug_embeddings).  Bar: bit-exact (CRC-32 of every 4096-byte row; the short prompt also in full)."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _crc_rows(t):
    b = t.contiguous().view(torch.int16).numpy()
    return np.array([zlib.crc32(b[i].tobytes()) for i in range(b.shape[0])], dtype=np.uint32)


def _engine_rows(text_ids, vocab, chunk, n_decode=0, pos_policy=0, want_tokens=None):
    """Runs one request with prefill chunks of `chunk` positions; returns ([T, 2048] bf16 = cond half | uncond half per position, rows of
    the first step, the decode rows [n_decode, 2048], the sampled speech ids)."""
    from chatterbox_vllm2_amd import engine as E
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors
    prompt = assemble_prompt_ids(text_ids); T = len(prompt)
    eng = E.T3Engine(n_layers=1, text_vocab=vocab, max_model_len=T + 16, max_seqs=1, kv_bytes=1 << 27, debug_logits=True, max_batched_rows=2 * chunk)
    eng.load_tensors(synthetic_tensors(1, vocab, 1234)); eng.finalize()
    eng.add_request(0, prompt, synthetic_cond_emb(1), E.make_sampling(temperature=0.0, repetition_penalty=1.0, max_tokens=n_decode + 1, ignore_eos=True, pos_policy=pos_policy))
    full = torch.zeros(T, 2048, dtype=torch.bfloat16); seen = np.zeros((T, 2), bool)
    first = None; dec = []
    while eng.num_unfinished():
        r = eng.step()
        rows, rs, rp = eng.debug_embeddings()
        assert rows.shape[0] == r.n_rows
        if r.n_prefill_rows:
            for i in range(rows.shape[0]):
                half = int(rs[i]) & 1                         # stream 2 * slot = conditional half, 2 * slot + 1 = unconditional half
                full[int(rp[i]), half * 1024:(half + 1) * 1024] = rows[i]; seen[int(rp[i]), half] = True
            if first is None:
                first = sorted(set(int(p) for p in rp))
        else:
            assert rows.shape[0] == 2 and int(rs[0]) == 0 and int(rs[1]) == 1 and int(rp[0]) == int(rp[1]) == T - 1 + len(dec) + 1
            dec.append(torch.cat([rows[0], rows[1]]))
    assert seen.all()
    ids = [t - 2500 for t in eng.get_output(0)[0]]
    eng.close()
    return full, first, (torch.stack(dec) if dec else None), ids


def test_full_prefill_blocks_match_reference_get_input_embeddings():
    z = np.load(os.path.join(G, "prompt_embeds.npz")); tok = json.load(open(os.path.join(G, "tokenizer.json")))
    for key, vocab in (("en_english_ids", 704), ("en_mtl_ids", 2454), ("es_mtl_ids", 2454)):
        full, _, _, _ = _engine_rows(tok[key], vocab, chunk=4096)
        assert np.array_equal(_crc_rows(full), z[f"full_{key}_crc"]), key
    full, _, _, _ = _engine_rows(z["short_text_ids"].tolist(), 2454, chunk=4096)
    assert np.array_equal(full.view(torch.int16).numpy(), z["short_full"])


def test_chunked_prefill_matches_reference_chunk_branches():
    """The reference covers a prefill block split over two steps (start-only chunk, end chunk with or without the conditioning's tail);
    here the scheduler's chunk size is set so that the FIRST step is exactly the reference's chunk A; the rest of the block (one or more
    steps here) must equal its chunk B."""
    z = np.load(os.path.join(G, "prompt_embeds.npz"))
    text = z["short_text_ids"].tolist(); T = len(z["short_ids"])
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    assert assemble_prompt_ids(text) == z["short_ids"].tolist()          # id layout built from the reference module's own constants
    for k in (1, 10, 33, 34, 35, 40, T - 1):
        full, first, _, _ = _engine_rows(text, 2454, chunk=k)
        assert first == list(range(k)), (k, first)
        crc = _crc_rows(full)
        assert np.array_equal(crc[:k], z[f"short_split{k}_a_crc"]), f"chunk A of split {k}"
        assert np.array_equal(crc[k:], z[f"short_split{k}_b_crc"]), f"chunk B of split {k}"


@pytest.mark.parametrize("policy", [0, 1])
def test_decode_rows_match_reference_decode_branch(policy):
    """pos_policy 1 = row [0, 0, :] of what the reference's decode branch literally returns for one token (index 0 for every decode
    token); pos_policy 0 = its row [0, k, :] (speech_pos_emb[k] for the k-th token: the exact per-sequence position)."""
    z = np.load(os.path.join(G, "prompt_embeds.npz"))
    _, _, dec, ids = _engine_rows(z["short_text_ids"].tolist(), 2454, chunk=4096, n_decode=4, pos_policy=policy)
    assert ids == z[f"short_greedy_ids_policy{policy}"].tolist()
    assert np.array_equal(_crc_rows(dec), z[f"short_decode_rows_policy{policy}_crc"])
