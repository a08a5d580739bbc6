"""Kernel-level parity: each HIP kernel, called through the C ABI (t3k_*), against the oracle's function
of the same name on the same seeded inputs.  Bar: BIT-EXACT (tolerance 0) -- the numerics contract
 fixes every rounding point and summation order."""
import numpy as np
import pytest
import torch

from util import assert_bit_equal, rand_bf16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from chatterbox_vllm2_amd import engine
    engine.load_library()
    return engine


def test_expf_bit_exact(E, oracle):
    x = torch.cat([torch.linspace(-90, 89, 20001), torch.randn(20000) * 10, torch.tensor([0.0, -0.0, -87.0, -87.0001, 88.0, 88.5, float("-inf")])])
    got = E.k_expf(x)
    want = torch.tensor([oracle.expf(float(v)) for v in x.tolist()], dtype=torch.float32)
    assert_bit_equal(got, want, "expf")
    # and the contract exp is a good exp: <= 2 ulp from the correctly rounded value on the softmax domain
    xs = x[(x > -80) & (x < 80)]
    ref = torch.exp(xs.double())
    rel = ((E.k_expf(xs).double() - ref).abs() / ref).max().item()
    assert rel < 3e-7, rel


@pytest.mark.parametrize("M,K,N,mt", [(2, 1024, 3072, 0), (1, 1024, 48, 1), (16, 1024, 1024, 0), (17, 1024, 64, 2),
                                        (64, 1024, 3072, 0), (64, 1024, 1024, 1), (64, 1024, 8194, 2),
                                        (200, 1024, 512, 2), (33, 1024, 16, 0)])      # the engine's 4-segment K (1024); other K: tools/legacy
def test_gemm_bit_exact(E, oracle, M, K, N, mt):
    x = rand_bf16(M, K, seed=M + K); W = rand_bf16(N, K, seed=N, scale=0.05)
    got = E.k_gemm(x, W, mt)
    want = oracle.gemm(x, W)
    assert_bit_equal(got, want, f"gemm {M}x{K}x{N} mt={mt}")
    # sanity against plain fp32 matmul (tolerance: reassociation error only)
    ref = x.float() @ W.float().T
    assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,K,N,mt", [(64, 1024, 1024, 0), (64, 4096, 1024, 0), (3, 4096, 48, 1), (130, 1024, 64, 4), (16, 4096, 32, 0), (2, 1024, 1024, 0)])
def test_gemm_16_segments_bit_exact(E, oracle, M, K, N, mt):
    """o_proj / down_proj form: 16 K-segments (16 waves) in one workgroup, groups of four folded ((G0+G1)+G2)+G3."""
    x = rand_bf16(M, K, seed=M + K + 1); W = rand_bf16(N, K, seed=N + 1, scale=0.05)
    assert_bit_equal(E.k_gemm(x, W, mt, nw=16), oracle.gemm(x, W, K // 16), f"16-segment gemm {M}x{K}x{N}")


@pytest.mark.parametrize("M,K", [(64, 1024), (5, 4096), (40, 4096), (81, 4096), (70, 1024), (96, 1024), (130, 1024), (129, 4096), (200, 1024), (256, 4096), (250, 4096)])  # 81+ rows: looped 16-wave schedule
def test_gemm_residual_epilogue_bit_exact(E, oracle, M, K):
    x = rand_bf16(M, K, seed=3); W = rand_bf16(1024, K, seed=4, scale=0.05); h = rand_bf16(M, 1024, seed=5, scale=2.0)
    y = oracle.gemm(x, W, K // 16).to(torch.bfloat16)
    want = (h.float() + y.float()).to(torch.bfloat16)
    assert_bit_equal(E.k_gemm_resid(x, W, h), want, "h + bf16(x W^T)")


@pytest.mark.parametrize("M,N", [(113, 1024), (545, 1024), (1100, 64), (3000, 64), (3071, 32)])
def test_down_proj_two_tiles_per_workgroup_bit_exact(E, oracle, M, N):
    """down_proj from 113 rows up to the prefill schedule's switch (3 072): gemm2_down2_kernel -- two n-tiles per sixteen-wave workgroup, the
    rows by LDS-DMA into each wave's own image, the next group's rows in flight under the fold.  Ragged row counts (a partial last group), one
    to many groups per workgroup (the narrow outputs give the launcher 128 / 256 row splits)."""
    x = rand_bf16(M, 4096, seed=M + 7); W = rand_bf16(N, 4096, seed=N + 4, scale=0.05); h = rand_bf16(M, N, seed=5, scale=2.0)
    y = oracle.gemm(x, W, 256).to(torch.bfloat16)
    assert_bit_equal(E.k_gemm_resid(x, W, h), (h.float() + y.float()).to(torch.bfloat16), f"down form M={M} N={N}")


@pytest.mark.parametrize("M,N", [(2, 3072), (64, 3072), (37, 8194), (200, 64)])
def test_norm_folded_gemm_bit_exact(E, oracle, M, N):
    """RMSNorm folded into the projection: rstd * GEMM(bf16(h * w_ln), W), statistic accumulated from the operand stream."""
    h = rand_bf16(M, 1024, seed=M, scale=3.0); ln = (rand_bf16(1024, seed=9) + 1.0).to(torch.bfloat16); W = rand_bf16(N, 1024, seed=N, scale=0.05)
    got = E.k_norm_gemm(h, ln, W)
    assert_bit_equal(got, oracle.norm_gemm(h, ln, W), f"norm+gemm {M}x{N}")
    # against the textbook formula in fp64 (tolerance: bf16 operand rounding)
    hd = h.double(); ref = (hd * torch.rsqrt(hd.pow(2).mean(-1, keepdim=True) + 1e-5) * ln.double()) @ W.double().T
    assert (got.double() - ref).abs().max().item() < 0.02 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M", [256, 257, 300, 515, 770])
def test_prefill_256x128_tiles_bit_exact(E, oracle, M, monkeypatch):
    """pgemm2_kernel (256 x 128 workgroup tiles, K staged 64 deep, 8 waves): every form of the prefill schedule against the same oracle
    functions, ragged row counts (a last row tile with 1 ... 255 rows); T3_PGEMM2_MIN_WGS=1 takes it at these small sizes (the engine
    switches to it where 192+ workgroups result)."""
    monkeypatch.setenv("T3_PGEMM2_MIN_WGS", "1"); monkeypatch.setenv("T3_PGEMM2_ALL", "1")
    E.k_set_prefill_rows(256, 0)
    try:
        _prefill_sized_checks(E, oracle, M)
    finally:
        E.k_set_prefill_rows(-1)


@pytest.mark.parametrize("M,wide", [(256, 0), (300, 0), (515, 0), (256, 256), (300, 256), (515, 256)])
def test_prefill_sized_gemms_bit_exact(E, oracle, M, wide, monkeypatch):
    """The LDS-tiled prefill schedule (pgemm_kernel, 128 x 64 or 128 x 128 workgroup tiles, row statistic in its own pass) must
    not change the numbers: every form against the same oracle functions, ragged row counts."""
    monkeypatch.setenv("T3_PGEMM2_MIN_WGS", "0")
    E.k_set_prefill_rows(256, wide)                        # the engine switches per form at 448-1600 rows; here the schedule is checked on small cases
    try:
        _prefill_sized_checks(E, oracle, M)
    finally:
        E.k_set_prefill_rows(-1)


def _prefill_sized_checks(E, oracle, M):
    x = rand_bf16(M, 1024, seed=M, scale=1.5); W = rand_bf16(128, 1024, seed=11, scale=0.05)
    assert_bit_equal(E.k_gemm(x, W), oracle.gemm(x, W), f"4-segment gemm M={M}")
    assert_bit_equal(E.k_gemm(x, W[:64], nw=16), oracle.gemm(x, W[:64], 64), f"16-segment gemm K=1024 M={M}")
    x4 = rand_bf16(M, 4096, seed=M + 1); W4 = rand_bf16(64, 4096, seed=12, scale=0.05)
    assert_bit_equal(E.k_gemm(x4, W4, nw=16), oracle.gemm(x4, W4, 256), f"16-segment gemm K=4096 M={M}")
    ln = (rand_bf16(1024, seed=9) + 1.0).to(torch.bfloat16)
    assert_bit_equal(E.k_norm_gemm(x, ln, W), oracle.norm_gemm(x, ln, W), f"norm+gemm M={M}")
    hres = rand_bf16(M, 64, seed=5, scale=2.0)
    y = oracle.gemm(x4, W4, 256).to(torch.bfloat16)
    assert_bit_equal(E.k_gemm_resid(x4, W4, hres), (hres.float() + y.float()).to(torch.bfloat16), f"residual epilogue M={M}")
    Wg = rand_bf16(64, 1024, seed=2, scale=0.1); Wu = rand_bf16(64, 1024, seed=3, scale=0.1)
    g = oracle.norm_gemm(x, ln, Wg).to(torch.bfloat16); u = oracle.norm_gemm(x, ln, Wu).to(torch.bfloat16)
    assert_bit_equal(E.k_silu_mul_gemm(x, ln, Wg, Wu), oracle.silu_mul(g, u), f"gate/up SiLU M={M}")


def test_norm_folded_gemm_row_gather(E, oracle):
    h = rand_bf16(50, 1024, seed=1, scale=2.0); ln = (rand_bf16(1024, seed=2) + 1.0).to(torch.bfloat16); W = rand_bf16(160, 1024, seed=3, scale=0.05)
    idx = [49, 0, 7, 7, 31, 12]
    assert_bit_equal(E.k_norm_gemm(h, ln, W, row_index=idx), oracle.norm_gemm(h[idx], ln, W), "gathered rows (speech-head form)")


def test_gemm_wide_dynamic_range(E, oracle):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(32, 1024, generator=g) * torch.exp2(torch.randint(-12, 12, (32, 1024), generator=g).float())).to(torch.bfloat16)
    W = (torch.randn(64, 1024, generator=g) * torch.exp2(torch.randint(-12, 12, (64, 1024), generator=g).float())).to(torch.bfloat16)
    assert_bit_equal(E.k_gemm(x, W), oracle.gemm(x, W), "gemm wide range")


@pytest.mark.parametrize("M", [2, 40, 97, 130, 256])        # 97+ rows: the weights-stationary looped schedule
def test_gate_up_silu_bit_exact(E, oracle, M):
    Fd = 512
    h = rand_bf16(M, 1024, seed=1, scale=2.0); ln = (rand_bf16(1024, seed=8) + 1.0).to(torch.bfloat16)
    Wg = rand_bf16(Fd, 1024, seed=2, scale=0.1); Wu = rand_bf16(Fd, 1024, seed=3, scale=0.1)
    got = E.k_silu_mul_gemm(h, ln, Wg, Wu)
    g = oracle.norm_gemm(h, ln, Wg).to(torch.bfloat16); u = oracle.norm_gemm(h, ln, Wu).to(torch.bfloat16)
    assert_bit_equal(got, oracle.silu_mul(g, u), "silu(gate)*up with folded norm")


@pytest.mark.parametrize("M", [81, 100, 129, 255, 256, 300, 447])
@pytest.mark.parametrize("pipe", ["1", "2", "0"])
def test_gate_up_pipelined_groups_bit_exact(E, oracle, M, pipe, monkeypatch):
    """gate/up at the model's width from 81 rows on (decode steps of 41+ utterances, C4's 256 rows, mixed steps up to 447): gemm2_pipe_kernel
    -- 16-row groups pipelined through compute waves (rows by LDS-DMA into a double-buffered swizzled image, weights stationary) and epilogue
    waves, one barrier per group -- and the serial walk it replaces (T3_GEMM_PIPE=0) against the oracle.  Row counts leave ragged last
    groups, odd group counts (uneven shares of the two workgroups of an n-group) and a single group for one of them."""
    monkeypatch.setenv("T3_GEMM_PIPE", pipe)
    if pipe != "1" and M not in (100, 256, 300):
        pytest.skip("the replaced form is checked at two row counts")
    Fd = 4096
    h = rand_bf16(M, 1024, seed=M, scale=2.0); ln = (rand_bf16(1024, seed=8) + 1.0).to(torch.bfloat16)
    Wg = rand_bf16(Fd, 1024, seed=2, scale=0.1); Wu = rand_bf16(Fd, 1024, seed=3, scale=0.1)
    got = E.k_silu_mul_gemm(h, ln, Wg, Wu)
    g = oracle.norm_gemm(h, ln, Wg).to(torch.bfloat16); u = oracle.norm_gemm(h, ln, Wu).to(torch.bfloat16)
    assert_bit_equal(got, oracle.silu_mul(g, u), f"silu(gate)*up, F=4096, M={M}, pipe={pipe}")


@pytest.mark.parametrize("M", [1100, 2047])
def test_weights_stationary_forms_up_to_their_new_switches(E, oracle, M):
    """Round 4 moved the switches to the LDS-tiled prefill schedule up (gate/up 448 -> 2 048 rows, qkv 704 -> 1 152, o 1 280 -> 2 048): the
    pipelined / looped forms at the row counts of continuous batching's mixed steps and beyond, ragged (M % 16 != 0), against the oracle.
    Narrower outputs than the model's keep the oracle affordable and give the launchers other workgroup splits (gate/up 32 n-groups x 8)."""
    h = rand_bf16(M, 1024, seed=M, scale=2.0); ln = (rand_bf16(1024, seed=8) + 1.0).to(torch.bfloat16)
    Wg = rand_bf16(1024, 1024, seed=2, scale=0.1); Wu = rand_bf16(1024, 1024, seed=3, scale=0.1)
    g = oracle.norm_gemm(h, ln, Wg).to(torch.bfloat16); u = oracle.norm_gemm(h, ln, Wu).to(torch.bfloat16)
    assert_bit_equal(E.k_silu_mul_gemm(h, ln, Wg, Wu), oracle.silu_mul(g, u), f"gate/up (pipelined) M={M}")
    x = rand_bf16(M, 1024, seed=M + 1); Wo = rand_bf16(1024, 1024, seed=4, scale=0.05); res = rand_bf16(M, 1024, seed=5, scale=2.0)
    y = oracle.gemm(x, Wo, 64).to(torch.bfloat16)
    assert_bit_equal(E.k_gemm_resid(x, Wo, res), (res.float() + y.float()).to(torch.bfloat16), f"o form (looped) M={M}")
    if M <= 1151:
        W = rand_bf16(3072, 1024, seed=21, scale=0.05)
        assert_bit_equal(E.k_qkv_gemm(h, ln, W), oracle.norm_gemm(h, ln, W).to(torch.bfloat16), f"qkv form (pipelined) M={M}")


@pytest.mark.parametrize("M", [2, 64, 128, 129, 200, 256, 300, 703])
def test_qkv_form_bit_exact(E, oracle, M):
    """The qkv projection as a step launches it (bf16 out, 3072 columns): the one-shot forms up to 128 rows, gemm2_pipe_kernel<3, BF16>
    (groups pipelined through compute and epilogue waves) from 129 rows up to the prefill schedule's threshold."""
    h = rand_bf16(M, 1024, seed=M, scale=2.0); ln = (rand_bf16(1024, seed=9) + 1.0).to(torch.bfloat16); W = rand_bf16(3072, 1024, seed=21, scale=0.05)
    assert_bit_equal(E.k_qkv_gemm(h, ln, W), oracle.norm_gemm(h, ln, W).to(torch.bfloat16), f"qkv form M={M}")


@pytest.mark.parametrize("M", [17, 20, 32, 33, 63, 64])
def test_gate_up_two_pairs_per_workgroup_bit_exact(E, oracle, M):
    """gate/up at the model's width (F = 4096) at 17-64 rows: two (gate, up) pairs per workgroup -- gemm2_kernel<1, 4, EPI_SILU> up to 48 rows,
    gemm2_pipe16_kernel from 49 (the form C3's 64-row steps run); ragged row counts leave a partial m-tile / a single group for one workgroup."""
    split = "-"
    Fd = 4096
    h = rand_bf16(M, 1024, seed=M, scale=2.0); ln = (rand_bf16(1024, seed=8) + 1.0).to(torch.bfloat16)
    Wg = rand_bf16(Fd, 1024, seed=2, scale=0.1); Wu = rand_bf16(Fd, 1024, seed=3, scale=0.1)
    got = E.k_silu_mul_gemm(h, ln, Wg, Wu)
    g = oracle.norm_gemm(h, ln, Wg).to(torch.bfloat16); u = oracle.norm_gemm(h, ln, Wu).to(torch.bfloat16)
    assert_bit_equal(got, oracle.silu_mul(g, u), f"silu(gate)*up, F=4096, M={M}, split={split}")


@pytest.mark.parametrize("M", [2, 16, 34, 50, 64, 256])
def test_speech_head_form_bit_exact(E, oracle, M):
    """The speech head as a decode step of 17-32 utterances launches it: 8 194 columns (513 n-tiles, packed to 516), rows gathered through an
    index, 2 x 4 tiles per 4-wave workgroup and two workgroups per CU (gemm2_kernel<2, 4, BF16, ..., EWV = 0>)."""
    big = rand_bf16(M + 9, 1024, seed=M, scale=2.0); ln = (rand_bf16(1024, seed=9) + 1.0).to(torch.bfloat16)
    W = rand_bf16(8194, 1024, seed=21, scale=0.05)
    idx = [(5 * i + 3) % (M + 9) for i in range(M)]
    want = oracle.norm_gemm(big[idx], ln, W).to(torch.bfloat16)
    assert_bit_equal(E.k_head_gemm(big, ln, W, idx), want, f"speech head form M={M}")


@pytest.mark.parametrize("M", [64, 130, 256])
def test_speech_head_without_gather_bit_exact(E, oracle, M):
    """A decode-only step samples every row in row order: the head reads its rows directly (no index) -- the one-shot form up to 128 rows,
    from 129 rows (decode steps of 65+ utterances, C4) gemm2_pipe_kernel<3, BF16> over the 516 packed tiles (172 workgroups, weights stationary,
    the last tile group overhanging the 8 194 columns)."""
    h = rand_bf16(M, 1024, seed=M + 3, scale=2.0); ln = (rand_bf16(1024, seed=9) + 1.0).to(torch.bfloat16)
    W = rand_bf16(8194, 1024, seed=21, scale=0.05)
    assert_bit_equal(E.k_head_gemm(h, ln, W, list(range(M))), oracle.norm_gemm(h, ln, W).to(torch.bfloat16), f"speech head, identity gather, M={M}")



@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8, 9, 14, 16])
def test_decode_gemms_at_few_rows(E, oracle, M):
    """Decode steps of 1-8 utterances: the one-tile GEMMs issue only the activation-row loads that hold rows (gemm2_kernel's AV forms:
    1, 2, 4 of 8 instructions at <= 2, 4, 8 rows for the 512-byte slices, 1 of 2 at <= 8 rows for the 128-byte ones).  Every form the
    step runs -- qkv, o + residual, gate/up, down + residual, the speech head over gathered rows -- against the same oracle functions."""
    h = rand_bf16(M, 1024, seed=M, scale=2.0); ln = (rand_bf16(1024, seed=9) + 1.0).to(torch.bfloat16)
    W = rand_bf16(3072, 1024, seed=21, scale=0.05)
    assert_bit_equal(E.k_norm_gemm(h, ln, W), oracle.norm_gemm(h, ln, W), f"qkv form M={M}")
    for n in (16, 48, 64):                                  # tile groups of 1, 3, 4
        assert_bit_equal(E.k_norm_gemm(h, ln, W[:n]), oracle.norm_gemm(h, ln, W[:n]), f"norm+gemm N={n} M={M}")
    Wg = rand_bf16(512, 1024, seed=2, scale=0.1); Wu = rand_bf16(512, 1024, seed=3, scale=0.1)
    g = oracle.norm_gemm(h, ln, Wg).to(torch.bfloat16); u = oracle.norm_gemm(h, ln, Wu).to(torch.bfloat16)
    assert_bit_equal(E.k_silu_mul_gemm(h, ln, Wg, Wu), oracle.silu_mul(g, u), f"gate/up M={M}")
    for K in (1024, 4096):
        x = rand_bf16(M, K, seed=M + K); Wo = rand_bf16(1024, K, seed=4, scale=0.05); res = rand_bf16(M, 1024, seed=5, scale=2.0)
        y = oracle.gemm(x, Wo, K // 16).to(torch.bfloat16)
        assert_bit_equal(E.k_gemm_resid(x, Wo, res), (res.float() + y.float()).to(torch.bfloat16), f"residual epilogue K={K} M={M}")
        assert_bit_equal(E.k_gemm(x, Wo[:64], nw=16), oracle.gemm(x, Wo[:64], K // 16), f"16-segment gemm K={K} M={M}")
    big = rand_bf16(40, 1024, seed=1, scale=2.0); idx = [(7 * i + 3) % 40 for i in range(M)]
    Wh = rand_bf16(160, 1024, seed=3, scale=0.05)
    assert_bit_equal(E.k_norm_gemm(big, ln, Wh, row_index=idx), oracle.norm_gemm(big[idx], ln, Wh), f"gathered rows M={M}")


def _oracle_rope_attention(oracle, qkv, row_stream, row_pos, n_streams, max_pos):
    cos_t, sin_t = oracle.rope_table(max_pos)
    pos = torch.tensor(row_pos, dtype=torch.int32)
    q = oracle.rope(qkv[:, :1024], pos, cos_t, sin_t)
    k = oracle.rope(qkv[:, 1024:2048], pos, cos_t, sin_t)
    v = qkv[:, 2048:].contiguous()
    K = torch.zeros(n_streams, max_pos, 1024, dtype=torch.bfloat16); Vv = torch.zeros_like(K)
    for r, (s, p) in enumerate(zip(row_stream, row_pos)):
        K[s, p] = k[r]; Vv[s, p] = v[r]
    out = torch.empty(len(row_pos), 1024, dtype=torch.bfloat16)
    for r, (s, p) in enumerate(zip(row_stream, row_pos)):
        for h in range(16):
            out[r, h * 64:(h + 1) * 64] = oracle.attn_row(q[r, h * 64:(h + 1) * 64], K[s, :p + 1, h * 64:(h + 1) * 64], Vv[s, :p + 1, h * 64:(h + 1) * 64])
    return out


@pytest.mark.parametrize("lens", [[1], [64], [65], [5, 130, 63, 200], [300, 2]])
def test_rope_paged_attention_bit_exact(E, oracle, lens):
    """Prefill-shaped call: every position of every stream is a row (ragged lengths, partial last chunk)."""
    row_stream, row_pos = [], []
    for s, L in enumerate(lens):
        row_stream += [s] * L; row_pos += list(range(L))
    qkv = rand_bf16(len(row_pos), 3072, seed=sum(lens))
    max_pos = max(lens) + 3
    got = E.k_rope_attention(qkv, row_stream, row_pos, len(lens), max_pos)
    want = _oracle_rope_attention(oracle, qkv, row_stream, row_pos, len(lens), max_pos)
    assert_bit_equal(got, want, f"attention lens={lens}")


@pytest.mark.parametrize("layout", ["split_runs", "tiny_streams", "unaligned_chunks"])
def test_rope_paged_attention_row_layouts(E, oracle, layout):
    """Row lists the 16-rows-per-workgroup schedule and the 8-token V groups must survive: a stream continued later in the same
    launch (two runs with other streams between them), many streams of 1-3 rows in one tile, runs that start off the 8-token grid."""
    runs = {"split_runs": [(0, 0, 10), (1, 0, 5), (0, 10, 30), (2, 0, 1), (3, 0, 2), (1, 5, 21), (0, 30, 97)],
            "tiny_streams": [(s, 0, 1 + s % 3) for s in range(40)],
            "unaligned_chunks": [(0, 0, 3), (1, 0, 13), (0, 3, 70), (1, 13, 66), (2, 0, 129)]}[layout]
    row_stream, row_pos = [], []
    for s_, p0, p1 in runs:
        row_stream += [s_] * (p1 - p0); row_pos += list(range(p0, p1))
    n_streams = max(row_stream) + 1; max_pos = max(row_pos) + 2
    qkv = rand_bf16(len(row_pos), 3072, seed=len(row_pos))
    got = E.k_rope_attention(qkv, row_stream, row_pos, n_streams, max_pos)
    want = _oracle_rope_attention(oracle, qkv, row_stream, row_pos, n_streams, max_pos)
    assert_bit_equal(got, want, f"attention layout={layout}")


def test_attention_peaky_scores(E, oracle):
    """Large-magnitude q/k: softmax saturates, exp underflows to 0 for most keys (rule-26 style forcing input)."""
    L = 150
    qkv = rand_bf16(L, 3072, seed=3, scale=6.0)
    got = E.k_rope_attention(qkv, [0] * L, list(range(L)), 1, L)
    want = _oracle_rope_attention(oracle, qkv, [0] * L, list(range(L)), 1, L)
    assert_bit_equal(got, want, "attention peaky")


def _oracle_decode_attention(oracle, ctx_qkv, new_qkv, ctx, max_pos):
    """oracle.rope + oracle.attn_row over the same inputs as E.k_decode_attention: per stream the rotated K / V of its content's
    context rows, then `steps` newest positions appended one by one."""
    cos_t, sin_t = oracle.rope_table(max_pos)
    steps, rows = new_qkv.shape[0], new_qkv.shape[1]
    n_content, content_rows = ctx_qkv.shape[0], ctx_qkv.shape[1]
    pos_all = torch.arange(content_rows, dtype=torch.int32)
    Kc = [oracle.rope(ctx_qkv[c, :, 1024:2048], pos_all, cos_t, sin_t) for c in range(n_content)]      # rotated once per content
    Vc = [ctx_qkv[c, :, 2048:].contiguous() for c in range(n_content)]
    out = torch.empty(steps, rows, 1024, dtype=torch.bfloat16); kvn = torch.empty(steps, rows, 2, 1024, dtype=torch.bfloat16)
    for r in range(rows):
        n = int(ctx[r]) - 1
        K = torch.cat([Kc[r % n_content][:n], torch.zeros(steps, 1024, dtype=torch.bfloat16)])
        V = torch.cat([Vc[r % n_content][:n], torch.zeros(steps, 1024, dtype=torch.bfloat16)])
        for s in range(steps):
            p = torch.tensor([n + s], dtype=torch.int32)
            q = oracle.rope(new_qkv[s, r:r + 1, :1024], p, cos_t, sin_t)[0]
            K[n + s] = oracle.rope(new_qkv[s, r:r + 1, 1024:2048], p, cos_t, sin_t)[0]
            V[n + s] = new_qkv[s, r, 2048:]
            kvn[s, r, 0] = K[n + s]; kvn[s, r, 1] = V[n + s]
            L = n + s + 1
            for h in range(16):
                sl = slice(h * 64, (h + 1) * 64)
                out[s, r, sl] = oracle.attn_row(q[sl], K[:L, sl], V[:L, sl])
    return out, kvn


DECODE_CTX = [63, 64, 65, 255, 256, 257, 511, 512, 513, 559, 767, 768, 999]


@pytest.mark.parametrize("rows,waves", [(2, 8), (2, 4), (9, 4), (9, 8), (64, 4), (33, 8)])
def test_fused_decode_attention_bit_exact(E, oracle, rows, waves):
    """THE headline kernel (attention_kernel<waves, nt, FUSE>: 51 % of the decode step's GPU time) against oracle.rope + oracle.attn_row
    at the contexts the bench times it at and beyond: every KV-block boundary (256 / 512 / 768), chunk boundaries (63 / 64 / 65),
    >= 3 chunk iterations per wave (4 waves from ctx 513), the longest context of max_model_len 1000; the rows of a launch carry
    DIFFERENT contexts (row r of launch j takes DECODE_CTX[(r + j) % 13]), two consecutive launches per case (the second reads the
    K / V the first wrote through the fused path), and the written K / V themselves are read back and compared."""
    max_pos = 1001
    ctx_qkv = rand_bf16(min(rows, 3), 998, 3072, seed=rows)
    launches = len(DECODE_CTX) if rows <= 2 else (6 if rows <= 9 else 2)        # 9 rows x 6 launches still meet every context of DECODE_CTX (index (r + j) % 13)
    for j in range(launches):
        ctx = [DECODE_CTX[(r + j) % len(DECODE_CTX)] for r in range(rows)]
        new_qkv = rand_bf16(2, rows, 3072, seed=1000 * rows + j)
        got, kv_got = E.k_decode_attention(ctx_qkv, new_qkv, ctx, max_pos, waves)
        want, kv_want = _oracle_decode_attention(oracle, ctx_qkv, new_qkv, ctx, max_pos)
        assert_bit_equal(kv_got, kv_want, f"newest K / V written by the fused kernel, rows={rows} waves={waves} launch {j}")
        assert_bit_equal(got, want, f"fused decode attention rows={rows} waves={waves} launch {j}")


def test_fused_decode_attention_long_contexts(E, oracle):
    """max_model_len goes up to 8192 (t3_create): the fused decode kernel at contexts of 2 049 .. 8 000 tokens (32-125 chunks per row, 8-31 KV
    blocks per stream, 8 and 4 waves), two launches each."""
    max_pos = 8192
    ctx_qkv = rand_bf16(2, 8000, 3072, seed=91)
    for rows, waves, ctx in ((2, 8, [8000, 4097]), (3, 4, [2049, 4095, 6400])):
        new_qkv = rand_bf16(2, rows, 3072, seed=92 + rows)
        got, kv_got = E.k_decode_attention(ctx_qkv, new_qkv, ctx, max_pos, waves)
        want, kv_want = _oracle_decode_attention(oracle, ctx_qkv, new_qkv, ctx, max_pos)
        assert_bit_equal(kv_got, kv_want, f"newest K / V at contexts {ctx}")
        assert_bit_equal(got, want, f"fused decode attention at contexts {ctx}")


def test_fused_decode_attention_256_rows(E, oracle):
    """A 128-utterance decode step (C4): 256 rows in one launch, contexts spread over the 13 boundary cases, as the engine picks the waves."""
    rows, max_pos = 256, 1001
    ctx_qkv = rand_bf16(2, 998, 3072, seed=77)
    ctx = [DECODE_CTX[(5 * r) % len(DECODE_CTX)] for r in range(rows)]
    new_qkv = rand_bf16(1, rows, 3072, seed=78)
    got, kv_got = E.k_decode_attention(ctx_qkv, new_qkv, ctx, max_pos, 0)
    want, kv_want = _oracle_decode_attention(oracle, ctx_qkv, new_qkv, ctx, max_pos)
    assert_bit_equal(kv_got, kv_want, "newest K / V, 256 rows")
    assert_bit_equal(got, want, "fused decode attention, 256 rows")


def test_prefill_attention_beyond_tile_lds(E, oracle):
    """A prompt whose context no longer fits the 16-row tile kernel's LDS (> 35 chunks = 2 240 tokens: 160 KiB per CU) must still
    prefill: launch_attention sends such rows to the per-row kernel.  One stream of 2 300 positions in one launch (every row's KV is
    written, then attention over all of them); a sample of rows is compared with the oracle, the last ones included."""
    L = 2300
    qkv = rand_bf16(L, 3072, seed=23)
    got = E.k_rope_attention(qkv, [0] * L, list(range(L)), 1, L + 1)
    cos_t, sin_t = oracle.rope_table(L + 1)
    pos = torch.arange(L, dtype=torch.int32)
    q = oracle.rope(qkv[:, :1024], pos, cos_t, sin_t); k = oracle.rope(qkv[:, 1024:2048], pos, cos_t, sin_t); v = qkv[:, 2048:].contiguous()
    for r in [0, 63, 64, 1000, 2175, 2176, 2239, 2240, 2241, 2298, 2299]:
        want = torch.cat([oracle.attn_row(q[r, h * 64:(h + 1) * 64], k[:r + 1, h * 64:(h + 1) * 64], v[:r + 1, h * 64:(h + 1) * 64]) for h in range(16)])
        assert_bit_equal(got[r], want, f"row {r} of a {L}-token prefill")


SAMPLING_CASES = [
    dict(temperature=0.0),
    dict(temperature=0.8, top_p=0.8, repetition_penalty=2.0),
    dict(temperature=0.8, top_p=1.0, repetition_penalty=2.0),
    dict(temperature=1.3, top_k=40),
    dict(temperature=0.7, top_k=5, top_p=0.5, min_p=0.05, repetition_penalty=1.2, presence_penalty=0.1, frequency_penalty=0.2),
    dict(temperature=0.5, min_p=0.2),
    dict(temperature=2.0, top_p=0.05),
]


@pytest.mark.parametrize("kw", SAMPLING_CASES)
def test_sampler_bit_exact(E, oracle, kw):
    g = torch.Generator().manual_seed(11)
    for trial in range(6):
        logits2 = (torch.randn(2, 8208, generator=g) * (0.7 + trial)).to(torch.bfloat16)
        counts = torch.zeros(8194, dtype=torch.int32)
        counts[torch.randint(0, 8194, (50,), generator=g)] = torch.randint(1, 4, (50,), generator=g).int()
        counts_gpu = counts.to(torch.uint16).clone()
        spo = oracle.make_sampling(seed=1234 + trial, uid=7 * trial, **kw)
        spe = E.make_sampling(seed=1234 + trial, uid=7 * trial, **kw)
        for step in (0, 1, 77):
            tok, lg = E.k_sample(logits2, counts_gpu.clone(), spe, 0.5, step)
            lc, lu = logits2[0, :8194].float(), logits2[1, :8194].float()
            cfg = (lc + (0.5 * (lc - lu).to(torch.bfloat16).float()).to(torch.bfloat16).float()).to(torch.bfloat16).float()
            assert_bit_equal(lg, cfg, "CFG logits")
            want = oracle.sample(cfg, counts, spo, step)
            assert tok == want, (kw, trial, step, tok, want)


def test_sampler_support_on_the_device(E, oracle):
    """VERDICT r3 item 4, device half: on 64 of the 2 000 random cases of tests/test_oracle.py::test_sampler_support_matches_vllm_mask_
    restatement the HIP sampler's SUPPORT (the ids its masks leave drawable, t3k_sample_support) equals the oracle's bit for bit and
    passes the same comparison with the independent restatement of vLLM's masks (tests/vllm_masks.py); the drawn id equals the oracle's."""
    from vllm_masks import check_support, sampler_case
    for i in range(0, 2000, 31)[:64]:
        lg, cnt, kw = sampler_case(i)
        l2 = torch.zeros(2, 8208); l2[0, :8194] = lg; l2[1, :8194] = lg          # cond == uncond: CFG returns the row itself (bf16-valued)
        tok, keep = E.k_sample_support(l2.to(torch.bfloat16), cnt.to(torch.uint16), E.make_sampling(seed=i, **kw), 0.5, i % 5)
        otok, okeep = oracle.sample_support(lg, cnt, oracle.make_sampling(seed=i, **kw), i % 5)
        assert tok == otok, (i, kw, tok, otok)
        assert torch.equal(keep, okeep), (i, kw, int(keep.sum()), int(okeep.sum()))
        check_support(i, lg, cnt, kw, tok, keep)


def test_sampler_ties_and_degenerate(E, oracle):
    """bf16 logits tie constantly; all-equal logits and a single dominant logit are the edge cases."""
    for fill, spike in ((0.0, None), (1.0, (4321, 30.0)), (-3.0, (0, 1.0))):
        l = torch.full((2, 8208), fill).to(torch.bfloat16)
        if spike:
            l[0, spike[0]] = spike[1]; l[1, spike[0]] = spike[1]
        for kw in (dict(temperature=0.0), dict(temperature=1.0, top_p=0.3), dict(temperature=1.0, top_k=3), dict(temperature=1.0)):
            counts = torch.zeros(8194, dtype=torch.uint16)
            tok, lg = E.k_sample(l, counts, E.make_sampling(seed=3, uid=1, **kw), 0.5, 5)
            want = oracle.sample(lg, torch.zeros(8194, dtype=torch.int32), oracle.make_sampling(seed=3, uid=1, **kw), 5)
            assert tok == want, (fill, spike, kw, tok, want)


def test_sampler_threshold_inside_tie_groups(E, oracle):
    """The nucleus threshold landing INSIDE a group of equal weights (few-valued logits: thousands of ties per value), at the largest weight, and
    the degenerate thresholds (top_p so small that nothing but one id survives; top_p combined with min-p / top-k, which re-sum the mass): the
    sampler's support and id against the oracle's, bit for bit.  Exercises the rank scan among ties (0 < dropped < ties), the no-scan shortcuts
    (none / all of the ties dropped), the descent that ends on a single-valued bin, and the counted-the-long-way path (threshold == total mass)."""
    g = torch.Generator().manual_seed(5)
    for levels, scale in ((2, 1.0), (5, 0.5), (16, 0.25), (200, 0.05)):
        vals = torch.randint(0, levels, (8194,), generator=g).float() * scale
        l2 = torch.zeros(2, 8208); l2[0, :8194] = vals; l2[1, :8194] = vals
        l2 = l2.to(torch.bfloat16); lg = l2[0, :8194].float()
        cnt = torch.zeros(8194, dtype=torch.int32)
        for kw in (dict(top_p=0.5), dict(top_p=0.9), dict(top_p=0.999), dict(top_p=0.013), dict(top_p=1e-9), dict(top_p=1e-30),
                   dict(top_p=0.7, min_p=0.3), dict(top_p=0.6, top_k=40), dict(top_p=0.95, top_k=3000, min_p=0.01)):
            kw = dict(temperature=1.0, **kw)
            tok, keep = E.k_sample_support(l2, cnt.to(torch.uint16), E.make_sampling(seed=9, uid=2, **kw), 0.5, 3)
            otok, okeep = oracle.sample_support(lg, cnt, oracle.make_sampling(seed=9, uid=2, **kw), 3)
            assert torch.equal(keep, okeep), (levels, kw, int(keep.sum()), int(okeep.sum()))
            assert tok == otok, (levels, kw, tok, otok)


def test_sampler_counts_update(E):
    l = torch.zeros(2, 8208).to(torch.bfloat16); l[:, 100] = 9.0
    counts = torch.zeros(8194, dtype=torch.uint16)
    tok, _ = E.k_sample(l, counts, E.make_sampling(temperature=0.0), 0.5, 0)
    assert tok == 100 and int(counts[100]) == 1 and int(counts.to(torch.int32).sum()) == 1


def test_handoff_kernel_matches_reference_analyzer_decisions(E):
    """f4: the device hand-off kernel against the 128 cases recorded from the reference's own AlignmentStreamAnalyzer
    (tests/golden/postfilter.json, make_golden.py g8) and against the host restatement t3_clean_tokens on the same inputs."""
    import json, os
    from chatterbox_vllm2_amd.postfilter import analyze_and_clean_tokens
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postfilter.json")))
    assert len(cases) >= 100
    for c in cases:
        kept, row = E.k_handoff(c["tokens"], c["text_token_count"], flags=0, ld=max(1, len(c["tokens"])) + 3)
        assert kept == c["cleaned"], (c["text_token_count"], len(c["tokens"]))
        assert all(v == 0 for v in row[len(kept):])                                     # padding
        want = [t for t in c["cleaned"] if 0 <= t < 6561]                               # tts.py:514
        assert E.k_handoff(c["tokens"], c["text_token_count"], flags=1)[0] == want
        assert want == analyze_and_clean_tokens(c["tokens"], c["text_token_count"], range_filter=True)[0]
    rs = np.random.RandomState(5)                                                       # long utterances: several 256-token sweeps
    for n, tc in ((1000, 400), (777, 600), (300, 2), (0, 5)):
        ids = rs.randint(0, 8194, size=n).tolist()
        assert E.k_handoff(ids, tc, flags=1)[0] == analyze_and_clean_tokens(ids, tc, range_filter=True)[0]
