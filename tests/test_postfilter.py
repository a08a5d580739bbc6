"""the host-side token post-filter (t3_clean_tokens) against decisions recorded from the REFERENCE's own
AlignmentStreamAnalyzer (tests/golden/postfilter.json, made by tests/golden/make_golden.py g8).  Integer logic: exact."""
import json
import os

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_clean_tokens_matches_reference_decisions():
    from chatterbox_vllm2_amd.postfilter import analyze_and_clean_tokens
    cases = json.load(open(os.path.join(G, "postfilter.json")))
    assert len(cases) > 100
    truncated = 0
    for c in cases:
        got, why = analyze_and_clean_tokens(c["tokens"], c["text_token_count"])
        assert got == c["cleaned"], (c["text_token_count"], len(c["tokens"]))
        if len(got) < len(c["tokens"]):
            truncated += 1
            assert why in ("repetition", "long_tail")
        else:
            assert why is None
    assert truncated > 20


def test_range_filter_and_edges():
    from chatterbox_vllm2_amd.postfilter import analyze_and_clean_tokens
    assert analyze_and_clean_tokens([], 10) == ([], None)
    assert analyze_and_clean_tokens([5, 6561, 7, 6562, -1, 8000, 9], 100, range_filter=True)[0] == [5, 7, 9]      # tts.py:514
    assert analyze_and_clean_tokens([1, 1, 1, 2], 100) == ([1, 1], "repetition")          # the third repeat is not kept
    got, why = analyze_and_clean_tokens(list(range(100)), 4)                               # complete at frame 2 -> tail at frame 12
    assert (len(got), why) == (11, "long_tail")
