"""An INDEPENDENT restatement of the masks vLLM's sampler applies, as documented for vllm==0.10.0 (V1 `Sampler`: `apply_penalties`
-> `apply_temperature` -> min-p -> `apply_top_k_top_p`), written against torch tensors in float64 and sharing no code with the
oracle (oracle/t3_oracle.c orc_sample) or the HIP sampler.  vLLM itself is absent here (pyproject.toml:29), so this file encodes
SURVEY.md A.5's "penalties -> temperature -> top-k/p order" plus the published semantics of each mask:

  penalties  (vllm/model_executor/layers/utils.py apply_penalties): for tokens already generated (the reference forwards no prompt
             ids that fall inside the speech vocabulary): logit > 0 ? logit / r : logit * r; then
             logit -= frequency * count; logit -= presence * (count > 0)
  temperature: logit / T  (T < 1e-5 = greedy: argmax)
  min-p      : p = softmax(logits); keep p >= min_p * max(p)
  top-k      : sort ascending; the k-th largest VALUE is the cut; logits strictly below it are masked (ties at the cut stay)
  top-p      : on what top-k left: probs of the ascending sort, cumulative sum; mask where cumsum <= 1 - top_p; never the last

What the restatement CANNOT decide and therefore reports instead of deciding (the test skips exactly these):
  * tokens whose decision sits within `margin` of a threshold (the oracle's weights are floor(exp(.) 2^32) of a <= 2 ulp exp: its
    masses differ from an exact softmax by ~1e-7 relative);
  * tokens whose probability is below 2^-31 of the largest one: the build's draw runs on integer weights floor(exp(l - max) 2^32)
    (DESIGN.md "Sampler"), so such a token has weight 0 and can never be drawn, whatever the masks say; all of them together hold
    < 2e-6 of the mass (8 194 x 2^-32).  A stated property of the build's sampler, not of vLLM's masks;
  * WHICH members of a group of exactly tied probabilities straddling the top-p cut are dropped: torch.sort's order among equal
    values is unspecified (vLLM inherits that); only the NUMBER dropped from the group is defined.
"""
import torch


def vllm_support(logits: torch.Tensor, counts: torch.Tensor, temperature, top_k, top_p, min_p, repetition, presence, frequency, margin=1e-5):
    """logits [V] (any float dtype), counts [V] int.  Returns (keep [V] bool, unsure [V] bool, tie_group [V] bool, n_keep_in_tie_group,
    below_resolution [V] bool).  keep is authoritative outside `unsure`, `tie_group` and `below_resolution`; inside tie_group exactly
    n_keep_in_tie_group members are kept; below_resolution = p < 2^-31 max(p) (the integer-weight draw cannot return those)."""
    x = logits.double().clone()
    c = counts.double()
    seen = c > 0
    if repetition != 1.0:
        x = torch.where(seen, torch.where(x > 0, x / repetition, x * repetition), x)
    x = x - frequency * c
    x = x - presence * seen.double()
    V = x.numel()
    none = torch.zeros(V, dtype=torch.bool)
    if temperature < 1e-5:
        keep = torch.zeros(V, dtype=torch.bool); keep[int(torch.argmax(x))] = True
        ties = x == x.max()
        return keep, none.clone(), (ties if int(ties.sum()) > 1 else none.clone()), 1, none.clone()
    x = x / temperature
    p = torch.softmax(x, 0)
    keep = torch.ones(V, dtype=torch.bool)
    below = p < p.max() * 2.0 ** -31                       # below the resolution of the integer-weight draw
    unsure = none.clone()
    if min_p > 0:
        thr = min_p * p.max()
        keep &= p >= thr
        unsure |= (p - thr).abs() <= margin * thr
    if 0 < top_k < V:
        xs = torch.where(keep, x, torch.full_like(x, float("-inf")))
        kth = torch.sort(xs, descending=False).values[V - top_k]
        keep &= xs >= kth                                  # strictly-below is masked; -inf >= -inf keeps nothing new (keep is and-ed)
        keep &= xs > float("-inf")
        # top-k compares LOGITS exactly (ties included): nothing is unsure here unless min-p made the candidate set unsure
    tie_group, n_tie_keep = none.clone(), 0
    if top_p < 1.0:
        xs = torch.where(keep, x, torch.full_like(x, float("-inf")))
        q = torch.softmax(xs, 0)
        qs, idx = torch.sort(q, descending=False, stable=True)
        cs = torch.cumsum(qs, 0)
        drop_sorted = cs <= (1.0 - top_p)
        drop_sorted[-1] = False
        drop = torch.zeros(V, dtype=torch.bool); drop[idx] = drop_sorted
        near = torch.zeros(V, dtype=torch.bool); near[idx] = (cs - (1.0 - top_p)).abs() <= margin
        unsure |= near & keep
        # exact ties straddling the cut
        n_drop = int(drop_sorted.sum())
        if 0 < n_drop < V:
            cut_val = qs[n_drop - 1]
            if qs[n_drop] == cut_val:                       # the first kept entry ties with the last dropped one
                grp = (q == cut_val) & keep
                tie_group = grp
                n_tie_keep = int((grp & ~drop).sum())
        keep &= ~drop
    return keep, unsure, tie_group, n_tie_keep, below


def sampler_case(i):
    """Random case i: bf16 logits at several scales (some coarsely quantised: exact ties), sparse counts, random parameters."""
    g = torch.Generator().manual_seed(9000 + i)
    V = 8194
    scale = (0.5, 2.0, 8.0, 30.0)[i % 4]
    lg = torch.randn(V, generator=g) * scale
    if i % 3 == 0:
        lg = torch.round(lg * 4) / 4                      # multiples of 0.25: many exact ties, also at the top
    if i % 7 == 0:
        lg[torch.randint(0, V, (3,), generator=g)] = lg.max()      # tied maxima
    lg = lg.to(torch.bfloat16).float()
    cnt = torch.zeros(V, dtype=torch.int32)
    n_seen = int(torch.randint(0, 120, (1,), generator=g))
    if n_seen:
        idx = torch.topk(lg + torch.randn(V, generator=g) * scale, n_seen).indices     # generated tokens are likely ones
        cnt[idx] = torch.randint(1, 6, (n_seen,), generator=g, dtype=torch.int32)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=g))]
    kw = dict(temperature=pick((0.0, 0.3, 0.8, 0.8, 1.0, 1.5)), top_k=pick((0, 0, 1, 5, 50, 1000)), top_p=pick((1.0, 0.95, 0.8, 0.8, 0.3)),
              min_p=pick((0.0, 0.0, 0.05, 0.3)), repetition_penalty=pick((1.0, 1.2, 2.0, 2.0)), presence_penalty=pick((0.0, 0.0, 0.5, -0.5)),
              frequency_penalty=pick((0.0, 0.0, 0.3)))
    return lg, cnt, kw


def check_support(i, lg, cnt, kw, tok, keep):
    """(drawn id, support) of a sampler under test against vllm_support on case i; returns the number of undecidable tokens skipped."""
    want, unsure, tie, n_tie, below = vllm_support(lg, cnt, kw["temperature"], kw["top_k"], kw["top_p"], kw["min_p"], kw["repetition_penalty"],
                                            kw["presence_penalty"], kw["frequency_penalty"])
    assert bool(keep[tok]), f"case {i} {kw}: the drawn id {tok} is outside the reported support"
    if kw["temperature"] < 1e-5:
        assert bool((tie if int(tie.sum()) else want)[tok]), f"case {i}: greedy id is not a maximum"
        if int(tie.sum()):
            assert tok == int(torch.nonzero(tie)[0]), f"case {i}: greedy must take the FIRST maximum"
        return 0
    firm = ~(unsure | tie | below)
    diff = (keep != want) & firm
    assert not bool(diff.any()), (f"case {i} {kw}: support differs from the vLLM restatement at ids {torch.nonzero(diff).flatten()[:8].tolist()} "
                                  f"(oracle keeps {int(keep.sum())}, restatement {int(want.sum())})")
    if int(tie.sum()) and not bool((unsure & tie).any()):
        assert int((keep & tie).sum()) == n_tie, f"case {i} {kw}: {int((keep & tie).sum())} of the tied group kept, restatement {n_tie}"
    return int((unsure | tie).sum())
