"""vLLM-shaped surface of the T3 engine: ``LLM`` / ``SamplingParams`` / ``RequestOutput``.

Drop-in for what src/chatterbox_vllm/tts.py uses of vLLM (reference lines):
  * construction  ``LLM(model=dir, task="generate", tokenizer="EnTokenizer"|"MtlTokenizer",
                        tokenizer_mode="custom", gpu_memory_utilization=f, enforce_eager=b,
                        max_model_len=n, **kwargs)``                              tts.py:150-171
  * generation    ``llm.generate([{"prompt": str, "multi_modal_data": {"conditionals": [Tensor[34,1024]]}}, ...],
                                 sampling_params=SamplingParams(temperature=, stop_token_ids=[9062],
                                 max_tokens=, top_p=, repetition_penalty=, **kw))``        tts.py:445-465
  * results       ``for r in results: for o in r.outputs: o.token_ids``  (ids >= 2500)     tts.py:474-492
Errors: bad input -> ValueError (HTTP 400 in api_server.py:323-326), anything else -> RuntimeError (500).
"""
from __future__ import annotations

import os
import warnings
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Union

import torch

from . import constants as C
from .engine import T3Engine, make_sampling
from .prompt import TextTokenizer, assemble_prompt_ids, find_tokenizer_file
from .weights import iter_safetensors, synthetic_tensors


class SamplingParams:
    """The subset of vllm.SamplingParams that reaches the sampler on this path (tts.py:455-464),
    with vLLM's names and defaults.  Unknown keyword arguments are rejected like vLLM does."""

    _IGNORED = {"logprobs", "prompt_logprobs", "detokenize", "skip_special_tokens", "spaces_between_special_tokens",
                "include_stop_str_in_output", "output_kind", "stop", "bad_words", "logit_bias", "allowed_token_ids",
                "truncate_prompt_tokens", "guided_decoding", "extra_args", "min_tokens", "best_of", "logits_processors"}

    def __init__(self, n: int = 1, temperature: float = 1.0, top_p: float = 1.0, top_k: int = 0, min_p: float = 0.0,
                 repetition_penalty: float = 1.0, presence_penalty: float = 0.0, frequency_penalty: float = 0.0,
                 seed: Optional[int] = None, stop_token_ids: Optional[List[int]] = None, max_tokens: Optional[int] = 16,
                 ignore_eos: bool = False, pos_policy: int = 0, **kwargs):
        for k, v in kwargs.items():
            if k not in self._IGNORED:
                raise TypeError(f"SamplingParams got an unexpected keyword argument {k!r}")
            if v not in (None, 0, False, [], {}, True) and k not in ("detokenize", "skip_special_tokens"):
                warnings.warn(f"SamplingParams.{k} is not supported on the T3 path and is ignored")
        if n != 1:
            raise ValueError("n must be 1 on the T3 path")
        if temperature < 0:
            raise ValueError(f"temperature must be non-negative, got {temperature}")
        if not 0 < top_p <= 1:
            raise ValueError(f"top_p must be in (0, 1], got {top_p}")
        if top_k < -1:
            raise ValueError(f"top_k must be -1/0 (disable) or positive, got {top_k}")
        if not 0 <= min_p <= 1:
            raise ValueError(f"min_p must be in [0, 1], got {min_p}")
        if not 0 < repetition_penalty <= 2 + 1e-6 and repetition_penalty <= 0:
            raise ValueError(f"repetition_penalty must be positive, got {repetition_penalty}")
        if max_tokens is not None and max_tokens < 1:
            raise ValueError(f"max_tokens must be at least 1, got {max_tokens}")
        self.n, self.temperature, self.top_p, self.top_k, self.min_p = n, temperature, top_p, top_k, min_p
        self.repetition_penalty, self.presence_penalty, self.frequency_penalty = repetition_penalty, presence_penalty, frequency_penalty
        self.seed, self.stop_token_ids, self.max_tokens, self.ignore_eos = seed, list(stop_token_ids or []), max_tokens, ignore_eos
        self.pos_policy = pos_policy

    def __repr__(self):
        return ("SamplingParams(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in (
            "temperature", "top_p", "top_k", "min_p", "repetition_penalty", "seed", "stop_token_ids", "max_tokens", "ignore_eos")) + ")")


@dataclass
class CompletionOutput:
    index: int
    text: str
    token_ids: List[int]
    cumulative_logprob: Optional[float] = None
    logprobs: Optional[Any] = None
    finish_reason: Optional[str] = None
    stop_reason: Union[int, str, None] = None


@dataclass
class RequestOutput:
    request_id: str
    prompt: Optional[str]
    prompt_token_ids: List[int]
    outputs: List[CompletionOutput]
    finished: bool = True
    metrics: Dict[str, float] = field(default_factory=dict)


PromptType = Union[str, Dict[str, Any]]


class LLM:
    def __init__(self, model: str, task: str = "generate", tokenizer: Optional[str] = None,
                 tokenizer_mode: Optional[str] = None, gpu_memory_utilization: float = 0.9,
                 enforce_eager: bool = True, max_model_len: int = 1000, max_num_seqs: int = 256,
                 max_num_batched_tokens: int = 0, seed: int = 0, load_format: str = "auto", dtype: str = "bfloat16",
                 device_id: Optional[int] = None, tokenizer_file: Optional[str] = None, num_hidden_layers: Optional[int] = None,
                 kv_cache_bytes: int = 0, debug_logits: bool = False, honor_enforce_eager: Optional[bool] = None, **kwargs):
        if task != "generate":
            raise ValueError("only task='generate' exists on the T3 path")
        if dtype not in ("bfloat16", "auto", torch.bfloat16):
            raise ValueError("the T3 engine computes in bf16 storage / fp32 accumulation only")
        for k in kwargs:
            warnings.warn(f"LLM(...) keyword {k!r} has no meaning for the T3 engine and is ignored")
        self.model_dir = model
        self.tokenizer_kind = tokenizer or "EnTokenizer"
        text_vocab = C.TEXT_VOCAB_EN if self.tokenizer_kind == "EnTokenizer" else C.TEXT_VOCAB_MTL   # t3.py:270
        n_layers = num_hidden_layers or C.N_LAYERS
        self.max_model_len, self.seed = int(max_model_len), int(seed)
        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.device_count() > 1 else 0
        if torch.cuda.is_available():
            torch.cuda.set_device(device_id)      # torch-side work of this process (RCCL collectives of dp.py, hand-off tensors) runs on the engine's GPU
        # enforce_eager: in vLLM it switches off CUDA-graph capture and torch.compile (start-up time and memory knobs; the
        # reference server always passes True, api_server.py:157 -> tts.py:156,163).  Here a decode step is replayed from a
        # hipGraph captured on first use -- no compile step, a few MB -- and token ids do not depend on it, so by default the
        # flag is accepted and NOT applied.  `honor_enforce_eager=True` (a kwarg of this class, forwarded by tts.py:171's **kwargs;
        # default: the T3_HONOR_ENFORCE_EAGER environment variable, else False) makes enforce_eager=True mean launch-by-launch
        # steps, as in vLLM (debugging, profiling: +43 % step time at B = 1).
        if honor_enforce_eager is None:
            honor_enforce_eager = os.environ.get("T3_HONOR_ENFORCE_EAGER", "0") == "1"
        self.honor_enforce_eager = bool(honor_enforce_eager)
        eager = bool(enforce_eager) and self.honor_enforce_eager
        self.engine = T3Engine(n_layers=n_layers, text_vocab=text_vocab, max_model_len=max_model_len, max_seqs=max_num_seqs,
                               device_id=device_id, kv_bytes=kv_cache_bytes, gpu_memory_utilization=gpu_memory_utilization,
                               enforce_eager=eager, debug_logits=debug_logits, max_batched_rows=max_num_batched_tokens)
        ckpt = os.path.join(model, "model.safetensors") if model and os.path.isdir(model) else model
        if load_format == "dummy" or not (ckpt and os.path.exists(ckpt)):
            if load_format != "dummy":
                raise ValueError(f"no checkpoint at {ckpt!r}: pass load_format='dummy' for seeded synthetic weights")
            self.engine.load_tensors(synthetic_tensors(n_layers, text_vocab, seed=1234))
        else:
            self.engine.load_tensors(iter_safetensors(ckpt))
        self.engine.finalize()
        tf = find_tokenizer_file(self.tokenizer_kind, model if model and os.path.isdir(model) else None, tokenizer_file)
        self.tokenizer = TextTokenizer(self.tokenizer_kind, tf) if tf else None
        self._next_id = 0
        self._next_uid = 0            # RNG stream of the next unseeded request (see generate)

    # -- vLLM API -------------------------------------------------------------------------------
    def get_tokenizer(self):
        return self.tokenizer

    def _text_ids(self, p: PromptType) -> (Optional[str], List[int], torch.Tensor):
        if isinstance(p, str):
            raise ValueError("T3 prompts need multi_modal_data={'conditionals': [Tensor[34,1024]]} (t3.py:207-210)")
        conds = (p.get("multi_modal_data") or {}).get("conditionals")
        if conds is None or len(conds) != 1:
            raise ValueError("exactly one conditional embedding is required for prefill (t3.py:208-209)")
        cond = conds[0]
        if cond.shape[0] != C.CONDITIONING_SIZE:
            raise ValueError("Conditionals must be CONDITIONING_SIZE tokens long (t3.py:210)")
        if "prompt_token_ids" in p:
            return p.get("prompt"), [int(t) for t in p["prompt_token_ids"]], cond
        if self.tokenizer is None:
            raise ValueError("no tokenizer file found: pass tokenizer_file=... / CHATTERBOX_TOKENIZER_DIR, or give prompt_token_ids")
        return p["prompt"], self.tokenizer.encode(p["prompt"]), cond

    def generate(self, prompts: Union[PromptType, Sequence[PromptType]], sampling_params: Optional[Union[SamplingParams, Sequence[SamplingParams]]] = None,
                 use_tqdm: bool = False, uid_base: Optional[int] = None, uids: Optional[Sequence[int]] = None,
                 keep_for_handoff: bool = False) -> List[RequestOutput]:
        """Randomness (the sampler draws from Philox keyed by (seed, uid, step), include/t3_engine.h):
          * no SamplingParams.seed (what tts.py:455-464 does): every request takes the next RNG stream of this LLM object, so the
            same text submitted twice gives two different utterances, while a fresh process replays the same sequence -- vLLM's
            behaviour with its global `seed=0` default;
          * SamplingParams.seed given: that request is reproducible wherever it sits -- alone, in a batch, in a data-parallel shard
            (vLLM's per-request generator): it draws from (seed, stream 0) whatever uids / uid_base say;
          * uids (one per prompt) or uid_base (request i uses uid_base + i) given -- the data-parallel launcher, dp.py: the stream
            of an UNSEEDED request is the utterance's GLOBAL index, so a sharded run emits the ids of the one-GPU run.
        keep_for_handoff: the requests stay in the engine (finished) until `handoff_tokens(outputs, ...)` hands their ids to the
        vocoder from device memory; without it they are released here, as the reference's flow expects."""
        if isinstance(prompts, (str, dict)):
            prompts = [prompts]
        if sampling_params is None:
            sampling_params = SamplingParams()
        sps = list(sampling_params) if isinstance(sampling_params, (list, tuple)) else [sampling_params] * len(prompts)
        if len(sps) != len(prompts):
            raise ValueError("The lengths of prompts and sampling_params must be the same.")
        if uids is not None and len(uids) != len(prompts):
            raise ValueError("The lengths of prompts and uids must be the same.")
        if uids is None and uid_base is not None:
            uids = [int(uid_base) + i for i in range(len(prompts))]
        metas = []
        queued: List[int] = []
        uid0 = self._next_uid
        # ids stay in device memory for the hand-off only when asked for; free buffers for this call's requests are made now, outside
        # the step loop (requests kept by an earlier call hold theirs until handoff_tokens releases them).  A caller that switched the
        # retention on itself (T3Engine.reserve_handoff) keeps it: this call touches the flag only when it needs it and puts it back.
        kept_before = getattr(self.engine, "keeps_device_ids", False)
        if keep_for_handoff:
            self.engine.reserve_handoff(len(prompts))
        try:
            self._queue(prompts, sps, uids, metas, queued)
        except Exception:
            for rid in queued:             # nothing of this call has run yet: leave the engine as it was found
                self.engine.abort(rid)
            self._next_uid = uid0
            raise
        try:
            self.engine.run_until_done()
        except Exception:
            for rid in queued:             # e.g. a request that can never be admitted (KV pool too small): drop the whole call
                try:
                    self.engine.abort(rid)
                except Exception:
                    pass
            raise
        finally:
            if keep_for_handoff and not kept_before:
                self.engine.reserve_handoff(0)      # finished requests keep the buffers they hold; no new ones are made
        outs = []
        for rid, text, final in metas:
            toks, fr = self.engine.get_output(rid)
            if not keep_for_handoff:
                self.engine.release(rid)
            reason = {1: "stop", 2: "length"}.get(fr)
            outs.append(RequestOutput(request_id=str(rid), prompt=text, prompt_token_ids=final,
                                      outputs=[CompletionOutput(index=0, text="", token_ids=toks, finish_reason=reason,
                                                                stop_reason=(toks[-1] if fr == 1 else None))]))
        return outs

    def handoff_tokens(self, outputs: Sequence[RequestOutput], text_token_counts: Sequence[int], range_filter: bool = True):
        """The T3 -> S3Gen hand-off for a whole batch (replaces the loop of tts.py:483-514): for outputs of
        `generate(..., keep_for_handoff=True)` returns (speech_tokens int32 [n, L], speech_token_lens int32 [n]) CUDA tensors --
        ids already un-offset, cut where the reference's AlignmentStreamAnalyzer would force EOS (text_token_counts as tts.py:496:
        `len(prompt.split()) * 2`), filtered to [0, 6561) -- and releases the requests."""
        rids = [int(o.request_id) for o in outputs]
        try:
            return self.engine.handoff_tokens(rids, text_token_counts, range_filter)
        finally:
            for r in rids:
                try:
                    self.engine.release(r)
                except Exception:
                    pass

    def _queue(self, prompts, sps, uids, metas, queued):
        for i, (p, sp) in enumerate(zip(prompts, sps)):
            text, tids, cond = self._text_ids(p)
            final = assemble_prompt_ids(tids)
            stop = -1
            if sp.stop_token_ids:
                if len(sp.stop_token_ids) > 1:
                    raise ValueError("the T3 engine supports one stop token id")
                stop = int(sp.stop_token_ids[0]) - C.SPEECH_TOKEN_OFFSET          # 9062 -> 6562 (tts.py:458)
            max_tokens = sp.max_tokens if sp.max_tokens is not None else self.max_model_len
            if sp.seed is not None:
                uid = 0                    # a seeded request draws from (seed, 0) wherever it runs: alone, in a batch or in a data-parallel shard
            elif uids is not None:
                uid = int(uids[i])
            else:
                uid = self._next_uid; self._next_uid += 1
            esp = make_sampling(temperature=sp.temperature, top_p=sp.top_p, min_p=sp.min_p, repetition_penalty=sp.repetition_penalty,
                                presence_penalty=sp.presence_penalty, frequency_penalty=sp.frequency_penalty,
                                top_k=max(0, sp.top_k), max_tokens=max_tokens, ignore_eos=sp.ignore_eos, stop_token=stop,
                                seed=self.seed if sp.seed is None else sp.seed, uid=uid, pos_policy=sp.pos_policy)
            rid = self._next_id; self._next_id += 1
            self.engine.add_request(rid, final, cond, esp)
            queued.append(rid)
            metas.append((rid, text, final))

    def shutdown(self):
        self.engine.close()

    def __del__(self):
        try:
            self.engine.close()
        except Exception:
            pass
