"""ctypes binding of libt3engine.so (include/t3_engine.h).

The HIP extension is the only compute path: if the shared library is missing or no GPU is visible,
construction raises -- there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import constants as C

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("T3_ENGINE_LIB") or os.path.join(CSRC, "libt3engine.so")     # override: diagnostic builds only

T3_OK, T3_E_INVALID, T3_E_DEVICE, T3_E_NOMEM, T3_E_STATE, T3_E_NOTFOUND = 0, -1, -2, -3, -4, -5


class T3EngineConfig(ct.Structure):
    _fields_ = [
        ("device_id", ct.c_int32), ("n_layers", ct.c_int32), ("text_vocab", ct.c_int32),
        ("max_model_len", ct.c_int32), ("max_seqs", ct.c_int32), ("max_batched_rows", ct.c_int32),
        ("kv_bytes", ct.c_int64), ("gpu_memory_utilization", ct.c_float), ("cfg_scale", ct.c_float),
        ("enforce_eager", ct.c_int32), ("debug_logits", ct.c_int32), ("n_groups", ct.c_int32), ("_pad", ct.c_int32),
    ]


class T3Sampling(ct.Structure):
    _fields_ = [
        ("temperature", ct.c_float), ("top_p", ct.c_float), ("min_p", ct.c_float),
        ("repetition_penalty", ct.c_float), ("presence_penalty", ct.c_float), ("frequency_penalty", ct.c_float),
        ("top_k", ct.c_int32), ("max_tokens", ct.c_int32), ("ignore_eos", ct.c_int32), ("stop_token", ct.c_int32),
        ("seed", ct.c_uint64), ("uid", ct.c_uint64), ("pos_policy", ct.c_int32), ("_pad", ct.c_int32),
    ]


class T3StepResult(ct.Structure):
    _fields_ = [
        ("n_rows", ct.c_int32), ("n_prefill_rows", ct.c_int32), ("n_sampled", ct.c_int32), ("n_finished", ct.c_int32),
        ("n_running", ct.c_int32), ("n_waiting", ct.c_int32), ("finished_ids", ct.c_int64 * 64),
    ]


class T3Stats(ct.Structure):
    _fields_ = [
        ("steps", ct.c_int64), ("decode_steps", ct.c_int64), ("tokens_generated", ct.c_int64),
        ("prefill_rows", ct.c_int64), ("decode_rows", ct.c_int64),
        ("gpu_ms_total", ct.c_double), ("gpu_ms_decode", ct.c_double), ("algo_bytes_decode", ct.c_double),
        ("sum_ctx_decode", ct.c_double),
        ("kv_blocks_total", ct.c_int64), ("kv_blocks_free", ct.c_int64), ("weight_bytes", ct.c_int64), ("finished_dropped", ct.c_int64),
    ]


# every symbol include/t3_engine.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "t3_create", "t3_destroy", "t3_last_error", "t3_load_tensor", "t3_finalize_weights", "t3_add_request",
    "t3_step", "t3_run_until_done", "t3_run_steps", "t3_num_unfinished", "t3_get_output", "t3_get_timing", "t3_release_request", "t3_abort_request", "t3_handoff_tokens", "t3_reserve_handoff", "t3_pop_finished", "t3_debug_embeddings", "t3k_handoff",
    "t3_clean_tokens", "t3_debug_logits", "t3_step_times", "t3_stats", "t3_reset_stats", "t3_set_profile", "t3_set_profile_kernel", "t3_kernel_ms",
    "t3k_gemm", "t3k_norm_gemm", "t3k_qkv_gemm", "t3k_head_gemm", "t3k_gemm_resid", "t3k_silu_mul_gemm", "t3k_rope_attention", "t3k_decode_attention", "t3k_sample", "t3k_sample_support", "t3k_expf",
    "t3_cond_create", "t3_cond_destroy", "t3_cond_last_error", "t3_cond_load_tensor", "t3_cond_encode", "t3_cond_emotion_row",
    "t3k_ce_layernorm", "t3k_ce_linear", "t3k_ce_attention", "t3k_set_prefill_rows", "t3k_set_prefill_wide_rows",
]

KERNEL_CLASSES = ["gemm_qkv", "gemm_o", "gemm_gateup", "gemm_down", "gemm_head", "attention",
                  "rope_kv", "embed", "sampler"]


def build_library(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-s", "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C chatterbox-vllm2_amd/csrc).  The T3 engine has no fallback path.")
    L = ct.CDLL(LIB_PATH)
    vp, i32, i64 = ct.c_void_p, ct.c_int32, ct.c_int64
    L.t3_create.argtypes = [ct.POINTER(T3EngineConfig), ct.POINTER(vp)]
    L.t3_destroy.argtypes = [vp]
    L.t3_last_error.argtypes = [vp]; L.t3_last_error.restype = ct.c_char_p
    L.t3_load_tensor.argtypes = [vp, ct.c_char_p, vp, i32, i32]
    L.t3_finalize_weights.argtypes = [vp]
    L.t3_add_request.argtypes = [vp, i64, vp, i32, vp, ct.POINTER(T3Sampling)]
    L.t3_step.argtypes = [vp, ct.POINTER(T3StepResult)]
    L.t3_run_until_done.argtypes = [vp]
    L.t3_run_steps.argtypes = [vp, i32, ct.POINTER(i32)]
    L.t3_num_unfinished.argtypes = [vp]
    L.t3_get_output.argtypes = [vp, i64, vp, ct.POINTER(i32), ct.POINTER(i32)]
    L.t3_get_timing.argtypes = [vp, i64, vp]
    L.t3_release_request.argtypes = [vp, i64]
    L.t3_abort_request.argtypes = [vp, i64]
    L.t3_handoff_tokens.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp]
    L.t3_reserve_handoff.argtypes = [vp, i32]
    L.t3_pop_finished.argtypes = [vp, vp, i32]
    L.t3_debug_embeddings.argtypes = [vp, vp, vp, vp, ct.POINTER(i32)]
    L.t3k_handoff.argtypes = [vp, i32, i32, i32, vp, i32, ct.POINTER(i32)]
    L.t3_clean_tokens.argtypes = [vp, i32, i32, i32, vp, ct.POINTER(i32)]
    L.t3_debug_logits.argtypes = [vp, i64, vp]
    L.t3_stats.argtypes = [vp, ct.POINTER(T3Stats)]
    L.t3_step_times.argtypes = [vp, vp, vp, i32]
    L.t3_reset_stats.argtypes = [vp]
    L.t3_set_profile.argtypes = [vp, i32]
    L.t3_set_profile_kernel.argtypes = [vp, ct.c_char_p]
    L.t3_kernel_ms.argtypes = [vp, ct.c_char_p, ct.POINTER(ct.c_double), ct.POINTER(i64)]
    L.t3k_gemm.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32]
    L.t3k_norm_gemm.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32]
    L.t3k_head_gemm.argtypes = [vp, vp, vp, i32, vp, i32, vp]
    L.t3k_qkv_gemm.argtypes = [vp, vp, vp, i32, vp]
    L.t3k_gemm_resid.argtypes = [vp, vp, i32, i32, i32, vp]
    L.t3k_silu_mul_gemm.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.t3k_rope_attention.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.t3k_decode_attention.argtypes = [vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp]
    L.t3k_sample.argtypes = [vp, i32, vp, ct.POINTER(T3Sampling), ct.c_float, ct.c_uint32, vp, vp]
    L.t3k_sample_support.argtypes = [vp, i32, vp, ct.POINTER(T3Sampling), ct.c_float, ct.c_uint32, vp, vp]
    L.t3k_expf.argtypes = [vp, vp, i32]
    L.t3_cond_create.argtypes = [i32, ct.POINTER(vp)]
    L.t3_cond_destroy.argtypes = [vp]
    L.t3_cond_last_error.restype = ct.c_char_p; L.t3_cond_last_error.argtypes = [vp]
    L.t3_cond_load_tensor.argtypes = [vp, ct.c_char_p, vp, i64]
    L.t3_cond_encode.argtypes = [vp, vp, vp, i32, ct.c_float, vp]
    L.t3_cond_emotion_row.argtypes = [vp, ct.c_float, vp]
    L.t3k_ce_layernorm.argtypes = [vp, vp, vp, vp, i32]
    L.t3k_ce_linear.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32]
    L.t3k_ce_attention.argtypes = [vp, vp, vp, vp, i32, i32]
    L.t3k_set_prefill_rows.argtypes = [i32]
    L.t3k_set_prefill_wide_rows.argtypes = [i32]
    for s in ABI_SYMBOLS:
        if s not in ("t3_last_error", "t3_cond_last_error"):
            getattr(L, s).restype = ct.c_int
    _lib = L
    return L


class T3Error(RuntimeError):
    pass


def _raise(code: int, msg: str):
    # ValueError surfaces as HTTP 400 in the reference server (api_server.py:323-326)
    if code == T3_E_INVALID:
        raise ValueError(msg)
    if code == T3_E_NOMEM:
        raise MemoryError(msg)
    raise T3Error(f"[{code}] {msg}")


def make_sampling(temperature=0.8, top_p=1.0, min_p=0.0, repetition_penalty=2.0, presence_penalty=0.0,
                  frequency_penalty=0.0, top_k=0, max_tokens=1000, ignore_eos=False,
                  stop_token=C.STOP_SPEECH_TOKEN, seed=0, uid=0, pos_policy=0) -> T3Sampling:
    return T3Sampling(float(temperature), float(top_p), float(min_p), float(repetition_penalty),
                      float(presence_penalty), float(frequency_penalty), int(top_k), int(max_tokens),
                      int(bool(ignore_eos)), int(stop_token), int(seed), int(uid), int(pos_policy), 0)


class T3Engine:
    """Thin object wrapper over one engine handle (one GPU)."""

    def __init__(self, n_layers: int = C.N_LAYERS, text_vocab: int = C.TEXT_VOCAB_EN, max_model_len: int = 1000,
                 max_seqs: int = 32, device_id: int = 0, kv_bytes: int = 0, gpu_memory_utilization: float = 0.9,
                 cfg_scale: Optional[float] = None, enforce_eager: bool = True, debug_logits: bool = False,
                 max_batched_rows: int = 0, n_groups: int = 0):
        self.lib = load_library()
        if cfg_scale is None:
            cfg_scale = float(os.environ.get("CHATTERBOX_CFG_SCALE", "0.5"))   # t3.py:296
        self.cfg = T3EngineConfig(device_id, n_layers, text_vocab, max_model_len, max_seqs, max_batched_rows,
                                  int(kv_bytes), float(gpu_memory_utilization), float(cfg_scale),
                                  int(enforce_eager), int(debug_logits), int(n_groups), 0)
        self.h = ct.c_void_p()
        rc = self.lib.t3_create(ct.byref(self.cfg), ct.byref(self.h))
        if rc:
            _raise(rc, self.lib.t3_last_error(None).decode())

    def _chk(self, rc: int):
        if rc:
            _raise(rc, self.lib.t3_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.t3_destroy(self.h)
            self.h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def load_tensors(self, tensors: Iterable[Tuple[str, torch.Tensor]], strict: bool = False) -> List[str]:
        """Feed (checkpoint name, tensor) pairs; unknown names are skipped like t3.py:316-319."""
        skipped = []
        for name, t in tensors:
            t = t.to(torch.bfloat16).contiguous()
            rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (1, t.numel())
            rc = self.lib.t3_load_tensor(self.h, name.encode(), ct.c_void_p(t.data_ptr()), rows, cols)
            if rc == T3_E_NOTFOUND and not strict:
                skipped.append(name)
                continue
            self._chk(rc)
        return skipped

    def finalize(self):
        self._chk(self.lib.t3_finalize_weights(self.h))

    # ---- requests
    def add_request(self, req_id: int, prompt_ids: Sequence[int], cond_emb: torch.Tensor, sp: T3Sampling):
        ids = np.ascontiguousarray(np.asarray(prompt_ids, dtype=np.int32))
        cond = cond_emb.detach().to("cpu", torch.float32).contiguous()
        if tuple(cond.shape) != (C.CONDITIONING_SIZE, C.HIDDEN):
            raise ValueError(f"conditionals must be [{C.CONDITIONING_SIZE}, {C.HIDDEN}], got {tuple(cond.shape)}")
        self._chk(self.lib.t3_add_request(self.h, int(req_id), ct.c_void_p(ids.ctypes.data), len(ids),
                                          ct.c_void_p(cond.data_ptr()), ct.byref(sp)))

    def step(self) -> T3StepResult:
        r = T3StepResult()
        self._chk(self.lib.t3_step(self.h, ct.byref(r)))
        return r

    def run_until_done(self):
        self._chk(self.lib.t3_run_until_done(self.h))

    def run_steps(self, n: int) -> int:
        done = ct.c_int32(0)
        self._chk(self.lib.t3_run_steps(self.h, int(n), ct.byref(done)))
        return int(done.value)

    def num_unfinished(self) -> int:
        return int(self.lib.t3_num_unfinished(self.h))

    def get_output(self, req_id: int) -> Tuple[List[int], int]:
        """(offset-space token ids >= 2500, finish_reason)"""
        cap = int(self.cfg.max_model_len)
        buf = np.zeros(cap, dtype=np.int32)
        n = ct.c_int32(cap); fr = ct.c_int32(0)
        self._chk(self.lib.t3_get_output(self.h, int(req_id), ct.c_void_p(buf.ctypes.data), ct.byref(n), ct.byref(fr)))
        return buf[: n.value].tolist(), int(fr.value)

    def timing(self, req_id: int) -> Tuple[float, float, float, float]:
        """(added, admitted, first token, finished) in seconds since the engine was created"""
        t = (ct.c_double * 4)()
        self._chk(self.lib.t3_get_timing(self.h, int(req_id), t))
        return tuple(float(x) for x in t)

    def release(self, req_id: int):
        self._chk(self.lib.t3_release_request(self.h, int(req_id)))

    def handoff_tokens(self, req_ids: Sequence[int], text_token_counts: Sequence[int], range_filter: bool = True):
        """f4: the finished utterances' ids, post-filtered (tts.py:300-365) and range-filtered (tts.py:514) ON THE DEVICE, as one padded
        batch: (speech_tokens int32 [n, L] cuda tensor, speech_token_lens int32 [n] cuda tensor).  The ids never visit the host."""
        n = len(req_ids)
        rid = np.ascontiguousarray(np.asarray(req_ids, dtype=np.int64)); tc = np.ascontiguousarray(np.asarray(text_token_counts, dtype=np.int32))
        if n == 0 or len(tc) != n:
            raise ValueError("handoff_tokens needs one text_token_count per request id")
        ld = 1
        for r in req_ids:
            cnt = ct.c_int32(0); fr = ct.c_int32(0)
            self._chk(self.lib.t3_get_output(self.h, int(r), None, ct.byref(cnt), ct.byref(fr)))
            ld = max(ld, int(cnt.value))
        dev = torch.device("cuda", int(self.cfg.device_id))
        toks = torch.empty(n, ld, dtype=torch.int32, device=dev); lens = torch.empty(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        self._chk(self.lib.t3_handoff_tokens(self.h, ct.c_void_p(rid.ctypes.data), n, ct.c_void_p(tc.ctypes.data), int(bool(range_filter)),
                                             ct.c_void_p(toks.data_ptr()), ld, ct.c_void_p(lens.data_ptr())))
        return toks, lens

    def reserve_handoff(self, n_requests: int):
        """Keep finished utterances' ids in device memory (n_requests > 0: buffers for that many are made now) or stop doing so (0)."""
        self._chk(self.lib.t3_reserve_handoff(self.h, int(n_requests)))
        self.keeps_device_ids = int(n_requests) > 0

    def pop_finished(self, cap: int = 4096) -> List[int]:
        """ids of the requests that finished since the last call, oldest first"""
        buf = np.zeros(max(1, cap), dtype=np.int64)
        n = int(self.lib.t3_pop_finished(self.h, ct.c_void_p(buf.ctypes.data), int(cap)))
        if n < 0:
            self._chk(n)
        return buf[:n].tolist()

    def debug_embeddings(self):
        """(rows [n, 1024] bf16, streams [n], positions [n]) of the most recent step's embedded input rows (debug_logits engines)."""
        n = ct.c_int32(0)
        self._chk(self.lib.t3_debug_embeddings(self.h, None, None, None, ct.byref(n)))
        rows = int(n.value)
        out = torch.empty(rows, C.HIDDEN, dtype=torch.bfloat16); rs = np.zeros(rows, dtype=np.int32); rp = np.zeros(rows, dtype=np.int32)
        n = ct.c_int32(rows)
        self._chk(self.lib.t3_debug_embeddings(self.h, ct.c_void_p(out.data_ptr()), ct.c_void_p(rs.ctypes.data), ct.c_void_p(rp.ctypes.data), ct.byref(n)))
        return out, rs, rp

    def abort(self, req_id: int):
        """Drop a request in any state (a waiting one leaves the queue, a running one frees its slot and KV blocks)."""
        self._chk(self.lib.t3_abort_request(self.h, int(req_id)))

    def debug_logits(self, req_id: int) -> torch.Tensor:
        out = torch.empty(C.SPEECH_VOCAB, dtype=torch.float32)
        self._chk(self.lib.t3_debug_logits(self.h, int(req_id), ct.c_void_p(out.data_ptr())))
        return out

    # ---- measurement
    def stats(self) -> T3Stats:
        s = T3Stats()
        self._chk(self.lib.t3_stats(self.h, ct.byref(s)))
        return s

    def reset_stats(self):
        self._chk(self.lib.t3_reset_stats(self.h))

    def step_times(self, cap: int = 16384):
        """(ms [n] float32, rows [n] int32) of the most recent steps since reset_stats, oldest first; rows < 0: the step carried prefill rows"""
        ms = np.zeros(max(1, cap), dtype=np.float32); rows = np.zeros(max(1, cap), dtype=np.int32)
        n = int(self.lib.t3_step_times(self.h, ct.c_void_p(ms.ctypes.data), ct.c_void_p(rows.ctypes.data), int(cap)))
        if n < 0:
            self._chk(n)
        return ms[:n], rows[:n]

    def set_profile(self, on: bool, only: Optional[str] = None):
        """HIP events around every kernel launch of decode-only steps; only = one kernel class (the rest runs undisturbed)."""
        self._chk(self.lib.t3_set_profile_kernel(self.h, only.encode() if only else None))
        self._chk(self.lib.t3_set_profile(self.h, int(on)))

    def kernel_ms(self, name: str) -> Tuple[float, int]:
        ms = ct.c_double(0); n = ct.c_int64(0)
        self._chk(self.lib.t3_kernel_ms(self.h, name.encode(), ct.byref(ms), ct.byref(n)))
        return float(ms.value), int(n.value)


# ---------------------------------------------------------------- kernel-level entry points (parity tests)
def _chk_k(rc: int, what: str):
    if rc == T3_E_DEVICE:
        raise T3Error(f"{what}: no usable HIP device / HIP error (the kernels have no CPU fallback)")
    if rc:
        raise ValueError(f"{what}: invalid argument ({rc})")


def _bf(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.bfloat16 and t.device.type == "cpu"
    return t.contiguous()


def k_gemm(x: torch.Tensor, W: torch.Tensor, mt: int = 0, nw: int = 4) -> torch.Tensor:
    """nw = K segments per workgroup: 4 (qkv / gate-up / head form) or 16 (o_proj / down_proj form)."""
    x, W = _bf(x), _bf(W)
    M, K = x.shape; N = W.shape[0]
    out = torch.empty(M, N, dtype=torch.float32)
    _chk_k(load_library().t3k_gemm(x.data_ptr(), W.data_ptr(), M, K, N, out.data_ptr(), mt, nw), "t3k_gemm")
    return out


def k_norm_gemm(h: torch.Tensor, ln_w: torch.Tensor, W: torch.Tensor, row_index=None) -> torch.Tensor:
    """RMSNorm folded into the projection; optional gather of rows of h."""
    h, ln_w, W = _bf(h), _bf(ln_w), _bf(W)
    ri = None if row_index is None else np.ascontiguousarray(np.asarray(row_index, dtype=np.int32))
    M = h.shape[0] if ri is None else len(ri)
    out = torch.empty(M, W.shape[0], dtype=torch.float32)
    _chk_k(load_library().t3k_norm_gemm(h.data_ptr(), ln_w.data_ptr(), W.data_ptr(), M, W.shape[0], out.data_ptr(),
                                        None if ri is None else ri.ctypes.data, h.shape[0]), "t3k_norm_gemm")
    return out


def k_qkv_gemm(h: torch.Tensor, ln_w: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """The qkv projection as a step launches it (norm folded, bf16 out) -> [M, 3072] bf16"""
    h, ln_w, W = _bf(h), _bf(ln_w), _bf(W)
    assert tuple(W.shape) == (3072, C.HIDDEN)
    out = torch.empty(h.shape[0], 3072, dtype=torch.bfloat16)
    _chk_k(load_library().t3k_qkv_gemm(h.data_ptr(), ln_w.data_ptr(), W.data_ptr(), h.shape[0], out.data_ptr()), "t3k_qkv_gemm")
    return out


def k_head_gemm(h: torch.Tensor, ln_w: torch.Tensor, W: torch.Tensor, row_index) -> torch.Tensor:
    """The speech head as a step launches it (gathered rows, bf16 logits) -> [M, 8194] bf16"""
    h, ln_w, W = _bf(h), _bf(ln_w), _bf(W)
    ri = np.ascontiguousarray(np.asarray(row_index, dtype=np.int32))
    assert tuple(W.shape) == (C.SPEECH_VOCAB, C.HIDDEN)
    out = torch.empty(len(ri), 8208, dtype=torch.bfloat16)
    _chk_k(load_library().t3k_head_gemm(h.data_ptr(), ln_w.data_ptr(), W.data_ptr(), len(ri), ri.ctypes.data, h.shape[0], out.data_ptr()), "t3k_head_gemm")
    return out[:, : C.SPEECH_VOCAB].contiguous()


def k_gemm_resid(x: torch.Tensor, W: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """returns bf16(h + bf16(x W^T)) computed by the o_proj / down_proj form."""
    x, W = _bf(x), _bf(W); h = _bf(h).clone()
    _chk_k(load_library().t3k_gemm_resid(x.data_ptr(), W.data_ptr(), x.shape[0], x.shape[1], W.shape[0], h.data_ptr()), "t3k_gemm_resid")
    return h


def k_silu_mul_gemm(h: torch.Tensor, ln_w: torch.Tensor, Wg: torch.Tensor, Wu: torch.Tensor) -> torch.Tensor:
    h, ln_w, Wg, Wu = _bf(h), _bf(ln_w), _bf(Wg), _bf(Wu)
    M, Fd = h.shape[0], Wg.shape[0]
    out = torch.empty(M, Fd, dtype=torch.bfloat16)
    _chk_k(load_library().t3k_silu_mul_gemm(h.data_ptr(), ln_w.data_ptr(), Wg.data_ptr(), Wu.data_ptr(), M, Fd, out.data_ptr()), "t3k_silu_mul_gemm")
    return out


def k_rope_attention(qkv: torch.Tensor, row_stream, row_pos, n_streams: int, max_pos: int) -> torch.Tensor:
    qkv = _bf(qkv)
    rs = np.ascontiguousarray(np.asarray(row_stream, dtype=np.int32)); rp = np.ascontiguousarray(np.asarray(row_pos, dtype=np.int32))
    out = torch.empty(qkv.shape[0], C.HIDDEN, dtype=torch.bfloat16)
    _chk_k(load_library().t3k_rope_attention(qkv.data_ptr(), rs.ctypes.data, rp.ctypes.data, qkv.shape[0], n_streams, max_pos,
                                             out.data_ptr()), "t3k_rope_attention")
    return out


def k_decode_attention(ctx_qkv: torch.Tensor, new_qkv: torch.Tensor, ctx, max_pos: int, waves: int = 0):
    """The fused decode attention kernel (RoPE + KV write + paged attention in one launch), `steps` consecutive launches.
    ctx_qkv [n_content, content_rows, 3072] bf16 pre-RoPE context rows (stream r takes content r % n_content, positions 0 .. ctx[r] - 2),
    new_qkv [steps, rows, 3072], ctx int list [rows] -> (out [steps, rows, 1024] bf16, kv_new [steps, rows, 2, 1024] bf16)."""
    ctx_qkv, new_qkv = _bf(ctx_qkv), _bf(new_qkv)
    cx = np.ascontiguousarray(np.asarray(ctx, dtype=np.int32))
    steps, rows = new_qkv.shape[0], new_qkv.shape[1]
    assert ctx_qkv.dim() == 3 and ctx_qkv.shape[2] == 3072 and new_qkv.shape[2] == 3072 and len(cx) == rows
    out = torch.empty(steps, rows, C.HIDDEN, dtype=torch.bfloat16); kvn = torch.empty(steps, rows, 2, C.HIDDEN, dtype=torch.bfloat16)
    _chk_k(load_library().t3k_decode_attention(ctx_qkv.data_ptr(), ctx_qkv.shape[0], ctx_qkv.shape[1], new_qkv.data_ptr(), cx.ctypes.data,
                                               rows, steps, int(max_pos), int(waves), out.data_ptr(), kvn.data_ptr()), "t3k_decode_attention")
    return out, kvn


def k_sample(logits2: torch.Tensor, counts: torch.Tensor, sp: T3Sampling, cfg: float, step: int):
    """logits2 [2, ld] bf16 (cond, uncond); counts uint16 [8194] (updated in place). -> (token, post-CFG logits)"""
    logits2 = _bf(logits2)
    assert counts.dtype == torch.uint16 and counts.numel() == C.SPEECH_VOCAB
    tok = ct.c_int32(0)
    lg = torch.empty(C.SPEECH_VOCAB, dtype=torch.float32)
    _chk_k(load_library().t3k_sample(logits2.data_ptr(), logits2.shape[1], counts.data_ptr(), ct.byref(sp), ct.c_float(cfg),
                                     ct.c_uint32(step), ct.byref(tok), lg.data_ptr()), "t3k_sample")
    return int(tok.value), lg


def k_sample_support(logits2: torch.Tensor, counts: torch.Tensor, sp: T3Sampling, cfg: float, step: int):
    """k_sample plus the support of the draw -> (token, keep [8194] bool): the ids the masks leave drawable"""
    logits2 = _bf(logits2)
    assert counts.dtype == torch.uint16 and counts.numel() == C.SPEECH_VOCAB
    tok = ct.c_int32(0)
    keep = torch.zeros(C.SPEECH_VOCAB, dtype=torch.uint8)
    _chk_k(load_library().t3k_sample_support(logits2.data_ptr(), logits2.shape[1], counts.data_ptr(), ct.byref(sp), ct.c_float(cfg),
                                             ct.c_uint32(step), ct.byref(tok), keep.data_ptr()), "t3k_sample_support")
    return int(tok.value), keep.bool()


def k_handoff(ids, text_token_count: int, flags: int = 1, ld: int = 0):
    """the hand-off kernel on one utterance's speech-space ids -> (kept ids list, padded row as list)"""
    a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32)); ld = ld or max(1, len(a))
    out = np.zeros(ld, dtype=np.int32); n = ct.c_int32(0)
    _chk_k(load_library().t3k_handoff(a.ctypes.data if len(a) else None, len(a), int(text_token_count), int(flags), out.ctypes.data, ld, ct.byref(n)), "t3k_handoff")
    return out[: n.value].tolist(), out.tolist()


def k_expf(x: torch.Tensor) -> torch.Tensor:
    x = x.to(torch.float32).contiguous()
    y = torch.empty_like(x)
    _chk_k(load_library().t3k_expf(x.data_ptr(), y.data_ptr(), x.numel()), "t3k_expf")
    return y


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.device.type == "cpu"
    return t.detach().to(torch.float32).contiguous()


def k_ce_layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    x, w, b = _f32(x), _f32(w), _f32(b)
    y = torch.empty_like(x)
    _chk_k(load_library().t3k_ce_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), x.shape[0]), "t3k_ce_layernorm")
    return y


def k_ce_linear(x: torch.Tensor, W: torch.Tensor, bias=None, resid=None) -> torch.Tensor:
    x, W = _f32(x), _f32(W)
    bias = _f32(bias) if bias is not None else None
    resid = _f32(resid) if resid is not None else None
    out = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32)
    _chk_k(load_library().t3k_ce_linear(x.data_ptr(), W.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        resid.data_ptr() if resid is not None else None, out.data_ptr(),
                                        x.shape[0], x.shape[1], W.shape[0]), "t3k_ce_linear")
    return out


def k_ce_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    q, k, v = _f32(q), _f32(k), _f32(v)
    out = torch.empty_like(q)
    _chk_k(load_library().t3k_ce_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.shape[0], k.shape[0]), "t3k_ce_attention")
    return out


def k_set_prefill_rows(rows: int, wide_rows: int = -1) -> None:
    """Process-wide: row count from which GEMM launches take the prefill schedule (< 0: default), and from which its 4-segment
    forms take 128 x 128 tiles (0: never, < 0: default)."""
    load_library().t3k_set_prefill_rows(int(rows))
    load_library().t3k_set_prefill_wide_rows(int(wide_rows))
