"""ctypes binding of the CPU oracle (oracle/t3_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package must never import this module.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libt3oracle.so")

V = 8194
D = 1024


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "t3_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class Sampling(ct.Structure):
    _fields_ = [
        ("temperature", ct.c_float), ("top_p", ct.c_float), ("min_p", ct.c_float),
        ("repetition_penalty", ct.c_float), ("presence_penalty", ct.c_float),
        ("frequency_penalty", ct.c_float),
        ("top_k", ct.c_int32), ("max_tokens", ct.c_int32), ("ignore_eos", ct.c_int32),
        ("stop_token", ct.c_int32),
        ("seed", ct.c_uint64), ("uid", ct.c_uint64),
        ("pos_policy", ct.c_int32), ("_pad", ct.c_int32),
    ]


def make_sampling(temperature=0.8, top_p=1.0, min_p=0.0, repetition_penalty=2.0, presence_penalty=0.0,
                  frequency_penalty=0.0, top_k=0, max_tokens=1000, ignore_eos=False, stop_token=6562,
                  seed=0, uid=0, pos_policy=0) -> Sampling:
    return Sampling(temperature, top_p, min_p, repetition_penalty, presence_penalty, frequency_penalty,
                    int(top_k), int(max_tokens), int(bool(ignore_eos)), int(stop_token), int(seed), int(uid),
                    int(pos_policy), 0)


_lib = None


def set_threads(n: int) -> int:
    """OpenMP thread count of the oracle's parallel loops (the GPU box reports 256 CPUs but grants a 16-core share)."""
    import ctypes.util
    try:
        ct.CDLL(ctypes.util.find_library("gomp") or "libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        os.environ["OMP_NUM_THREADS"] = str(n)
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        L = ct.CDLL(_SO)
        vp, i32, f32 = ct.c_void_p, ct.c_int, ct.c_float
        L.orc_expf.restype = f32; L.orc_expf.argtypes = [f32]
        L.orc_gemm_nk.argtypes = [vp, vp, i32, i32, i32, vp, i32]
        L.orc_norm_gemm_nk.argtypes = [vp, vp, vp, i32, i32, vp]
        L.orc_row_rstd.argtypes = [vp, i32, vp]
        L.orc_rope_table.argtypes = [i32, vp, vp]
        L.orc_rope.argtypes = [vp, vp, i32, vp, vp]
        L.orc_attn_row.argtypes = [vp, vp, vp, i32, i32, vp]
        L.orc_silu_mul.argtypes = [vp, vp, vp, i32]
        L.orc_create.restype = vp; L.orc_create.argtypes = [i32, i32, i32, i32]
        L.orc_set_tensor.restype = i32; L.orc_set_tensor.argtypes = [vp, ct.c_char_p, vp, i32, i32]
        L.orc_destroy.argtypes = [vp]
        L.orc_forward_rows.argtypes = [vp, vp, vp, vp, i32, i32, vp]
        L.orc_cfg_logits.argtypes = [vp, vp, vp, f32, vp, vp, vp]
        L.orc_philox.argtypes = [ct.c_uint32] * 6 + [vp]
        L.orc_sample.restype = i32; L.orc_sample.argtypes = [vp, vp, ct.POINTER(Sampling), ct.c_uint32]
        L.orc_sample_support.restype = i32; L.orc_sample_support.argtypes = [vp, vp, ct.POINTER(Sampling), ct.c_uint32, vp]
        L.orc_prompt_embeds.restype = i32; L.orc_prompt_embeds.argtypes = [vp, vp, i32, vp, vp, vp]
        L.orc_decode_embed.restype = i32; L.orc_decode_embed.argtypes = [vp, i32, i32, vp]
        L.orc_generate.restype = i32
        L.orc_generate.argtypes = [vp, i32, vp, i32, vp, ct.POINTER(Sampling), f32, i32, vp, vp]
        L.orc_decode_steps_timing.argtypes = [vp, i32, i32, i32]
        L.orc_ce_layernorm.argtypes = [vp, vp, vp, vp, i32]
        L.orc_ce_linear.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32]
        L.orc_ce_attention.restype = i32; L.orc_ce_attention.argtypes = [vp, vp, vp, vp, i32, i32]
        L.orc_cond_enc.restype = i32; L.orc_cond_enc.argtypes = [vp, vp, vp, i32, f32, vp]
        _lib = L
    return _lib


def _bf(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.bfloat16 and t.device.type == "cpu"
    return t.contiguous()


def _p(t) -> int:
    return t.data_ptr() if isinstance(t, torch.Tensor) else t.ctypes.data


# ---------------------------------------------------------------- op-level entry points
def expf(x: float) -> float:
    return float(lib().orc_expf(ct.c_float(x)))


def gemm(x: torch.Tensor, W: torch.Tensor, seg_len: int = 0) -> torch.Tensor:
    """x [M,K] bf16, W [N,K] bf16 -> [M,N] fp32 in the contract summation order.
    seg_len: MFMA chain length per segment (K/seg_len must be 4 or 16); default K/4."""
    x, W = _bf(x), _bf(W)
    M, K = x.shape; N = W.shape[0]
    seg_len = seg_len or K // 4
    assert K % seg_len == 0 and K // seg_len in (4, 16) and seg_len % 32 == 0
    out = torch.empty(M, N, dtype=torch.float32)
    lib().orc_gemm_nk(_p(x), _p(W), M, K, N, _p(out), seg_len)
    return out


def norm_gemm(h: torch.Tensor, w: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """RMSNorm folded into the GEMM: rstd[m] * GEMM(bf16(h*w), W); h [M,1024], w [1024], W [N,1024] bf16 -> fp32 [M,N]."""
    h, w, W = _bf(h), _bf(w), _bf(W)
    out = torch.empty(h.shape[0], W.shape[0], dtype=torch.float32)
    lib().orc_norm_gemm_nk(_p(h), _p(w), _p(W), h.shape[0], W.shape[0], _p(out))
    return out


def row_rstd(h: torch.Tensor) -> torch.Tensor:
    h = _bf(h); out = torch.empty(h.shape[0], dtype=torch.float32)
    lib().orc_row_rstd(_p(h), h.shape[0], _p(out))
    return out


def rope_table(max_pos: int):
    c = torch.empty(max_pos, 32, dtype=torch.float32); s = torch.empty(max_pos, 32, dtype=torch.float32)
    lib().orc_rope_table(max_pos, _p(c), _p(s))
    return c, s


def rope(x: torch.Tensor, pos: torch.Tensor, cos_t, sin_t) -> torch.Tensor:
    """x [rows,1024] bf16 (16 heads x 64), pos int32 [rows]; returns rotated copy."""
    y = _bf(x).clone(); pos = pos.to(torch.int32).contiguous()
    lib().orc_rope(_p(y), _p(pos), y.shape[0], _p(cos_t), _p(sin_t))
    return y


def attn_row(q: torch.Tensor, K: torch.Tensor, Vv: torch.Tensor) -> torch.Tensor:
    """q [64] bf16; K, V [L,64] bf16 contiguous -> [64] bf16."""
    q, K, Vv = _bf(q), _bf(K), _bf(Vv)
    out = torch.empty(64, dtype=torch.bfloat16)
    lib().orc_attn_row(_p(q), _p(K), _p(Vv), K.shape[0], 64, _p(out))
    return out


def silu_mul(g: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    g, u = _bf(g), _bf(u)
    out = torch.empty_like(g)
    lib().orc_silu_mul(_p(g), _p(u), _p(out), g.numel())
    return out


def philox(c, k):
    out = (ct.c_uint32 * 4)()
    lib().orc_philox(c[0], c[1], c[2], c[3], k[0], k[1], out)
    return list(out)


def sample(logits: torch.Tensor, counts: torch.Tensor, sp: Sampling, step: int) -> int:
    logits = logits.to(torch.float32).contiguous(); counts = counts.to(torch.uint16).contiguous()
    return int(lib().orc_sample(_p(logits), _p(counts), ct.byref(sp), step))


def sample_support(logits: torch.Tensor, counts: torch.Tensor, sp: Sampling, step: int = 0):
    """(token, keep [8194] bool): the draw and the set of ids the draw can return (what the masks leave)."""
    logits = logits.to(torch.float32).contiguous(); counts = counts.to(torch.uint16).contiguous()
    keep = torch.zeros(8194, dtype=torch.uint8)
    tok = int(lib().orc_sample_support(_p(logits), _p(counts), ct.byref(sp), step, _p(keep)))
    return tok, keep.bool()


# ---------------------------------------------------------------- model-level
CE_ORDER = ("spkr_enc.weight", "spkr_enc.bias", "emotion_adv_fc.weight", "perceiver.pre_attention_query",
            "perceiver.attn.norm.weight", "perceiver.attn.norm.bias",
            "perceiver.attn.to_q.weight", "perceiver.attn.to_q.bias", "perceiver.attn.to_k.weight", "perceiver.attn.to_k.bias",
            "perceiver.attn.to_v.weight", "perceiver.attn.to_v.bias", "perceiver.attn.proj_out.weight", "perceiver.attn.proj_out.bias")


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def ce_layernorm(x, w, b):
    x = _f32(x); y = torch.empty_like(x)
    lib().orc_ce_layernorm(_p(x), _p(_f32(w)), _p(_f32(b)), _p(y), x.shape[0])
    return y


def ce_linear(x, W, bias=None, resid=None):
    x = _f32(x); W = _f32(W)
    out = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32)
    bias = _f32(bias) if bias is not None else None; resid = _f32(resid) if resid is not None else None
    lib().orc_ce_linear(_p(x), _p(W), _p(bias) if bias is not None else None, _p(resid) if resid is not None else None, _p(out),
                        x.shape[0], x.shape[1], W.shape[0])
    return out


def ce_attention(q, k, v):
    q, k, v = _f32(q), _f32(k), _f32(v)
    out = torch.empty_like(q)
    rc = lib().orc_ce_attention(_p(q), _p(k), _p(v), _p(out), q.shape[0], k.shape[0])
    if rc:
        raise ValueError("ce_attention: key count out of range")
    return out


def cond_enc(params: dict, speaker_emb, prompt_emb, emotion: float) -> torch.Tensor:
    """params: {"cond_enc.<name>": fp32 tensor} -> [34, 1024] fp32 (the reference's T3CondEnc.forward, cond_enc.py:80-123)."""
    keep = [_f32(params["cond_enc." + n]) for n in CE_ORDER]
    arr = (ct.c_void_p * len(keep))(*[_p(t) for t in keep])
    spk = _f32(speaker_emb).reshape(-1); pe = _f32(prompt_emb)
    out = torch.empty(34, 1024, dtype=torch.float32)
    rc = lib().orc_cond_enc(arr, _p(spk), _p(pe), pe.shape[0], ct.c_float(emotion), _p(out))
    if rc:
        raise ValueError("cond_enc: prompt length out of range")
    return out


class OracleModel:
    def __init__(self, n_layers: int, text_vocab: int, max_pos: int = 1024, n_streams: int = 2):
        self.n_layers, self.text_vocab, self.max_pos, self.n_streams = n_layers, text_vocab, max_pos, n_streams
        self.h = lib().orc_create(n_layers, text_vocab, max_pos, n_streams)

    def load(self, tensors):
        for name, t in tensors:
            t = _bf(t.to(torch.bfloat16))
            rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (1, t.shape[0])
            lib().orc_set_tensor(self.h, name.encode(), _p(t), rows, cols)
        return self

    def close(self):
        if self.h:
            lib().orc_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prompt_embeds(self, prompt_ids, cond_emb: torch.Tensor):
        ids = np.asarray(prompt_ids, dtype=np.int32); T = len(ids)
        cond = cond_emb.to(torch.float32).contiguous()
        ec = torch.empty(T, D, dtype=torch.bfloat16); eu = torch.empty(T, D, dtype=torch.bfloat16)
        rc = lib().orc_prompt_embeds(self.h, _p(ids), T, _p(cond), _p(ec), _p(eu))
        if rc:
            raise ValueError(f"orc_prompt_embeds failed: {rc}")
        return ec, eu

    def decode_embed(self, tok: int, k: int) -> torch.Tensor:
        """speech_emb[tok] + speech_pos_emb[k] as a bf16 row (both CFG halves get the same row, t3.py:480)"""
        out = torch.empty(D, dtype=torch.bfloat16)
        if lib().orc_decode_embed(self.h, int(tok), int(k), _p(out)):
            raise ValueError("decode_embed: token or position out of range")
        return out

    def forward_rows(self, h: torch.Tensor, row_stream, row_pos, tap_layer: int = -1):
        """Runs all layers in place on a copy of h ([rows,1024] bf16); returns (h_out, tap)."""
        h = _bf(h).clone()
        rs = np.asarray(row_stream, dtype=np.int32); rp = np.asarray(row_pos, dtype=np.int32)
        tap = torch.empty_like(h) if tap_layer >= 0 else None
        lib().orc_forward_rows(self.h, _p(h), _p(rs), _p(rp), h.shape[0], tap_layer, _p(tap) if tap is not None else None)
        return h, tap

    def cfg_logits(self, hc: torch.Tensor, hu: torch.Tensor, cfg: float, parts: bool = False):
        out = torch.empty(V, dtype=torch.float32)
        oc = torch.empty(V, dtype=torch.float32) if parts else None
        ou = torch.empty(V, dtype=torch.float32) if parts else None
        lib().orc_cfg_logits(self.h, _p(_bf(hc)), _p(_bf(hu)), ct.c_float(cfg), _p(out),
                             _p(oc) if parts else None, _p(ou) if parts else None)
        return (out, oc, ou) if parts else out

    def generate(self, prompt_ids, cond_emb: torch.Tensor, sp: Sampling, cfg: float = 0.5,
                 max_model_len: int = 1000, slot: int = 0, want_logits: bool = False):
        """Returns (speech-space ids list, logits [n,8194] or None)."""
        ids = np.asarray(prompt_ids, dtype=np.int32); T = len(ids)
        cond = cond_emb.to(torch.float32).contiguous()
        cap = max(1, min(sp.max_tokens, max_model_len - T))
        out = np.zeros(cap, dtype=np.int32)
        lg = torch.empty(cap, V, dtype=torch.float32) if want_logits else None
        n = lib().orc_generate(self.h, slot, _p(ids), T, _p(cond), ct.byref(sp), ct.c_float(cfg),
                               max_model_len, _p(out), _p(lg) if want_logits else None)
        if n < 0:
            raise ValueError(f"orc_generate failed: {n}")
        return out[:n].tolist(), (lg[:n] if want_logits else None)

    def decode_steps_timing(self, B: int, ctx: int, steps: int):
        lib().orc_decode_steps_timing(self.h, B, ctx, steps)
