#!/usr/bin/env python3
"""bench.py -- T3 decode throughput on MI355X (BASELINE.json metric: speech-tokens/s/GPU, batch 32, + p50 RTF).

Default workload (config.workload = "C3", BASELINE.json configs[2], the one the metric is quoted on): multilingual vocab 2454,
B = 32 utterances per GPU = 16 en prompts (T = 116) + 16 es prompts (T = 141), max_model_len = 1000, real layer count (30), bf16
weights, seeded synthetic weights (no checkpoint offline), the reference's sampling defaults (temperature 0.8, top-p 0.8, repetition
penalty 2.0, tts.py:377,416) with the stop id masked (fixed-length generation).
A "step" = one pass of the hot path over the batch = one decode step of all 32 utterances (64 CFG rows).
Prefill, fast-forward and warmup steps are outside the timed region; inputs (weights, prompts, KV) are resident in HBM.

The K timed steps are CENTRED on the middle of the utterances' lives (decode step (max_model_len - longest prompt) / 2, where the
context equals the whole-run mean, ~560 tokens for C3) whatever K is: the engine is fast-forwarded with untimed decode steps
first, so `--steps 20` and `--steps 800` measure the same operating point (config.ctx_first / ctx_last say which).
`e2e` (same JSON line) is the figure SURVEY.md 8(d) asks for beside it: the 32 utterances from `add_request` to their last token
(prefill + all 884 / 859 decode steps), tokens / wall, with the per-request RTF distribution (the reference's own clock is the wall
time of its generate call, tts.py:444,466-467; RTF_i = (finished - admitted) / (n_tokens_i / 25)).

`--workload c4` (BASELINE.json configs[3] and, with --gpus 8, configs[4]): the committed request stream tests/golden/c4_requests.json
(499 sentences of the reference's docs/benchmark-text-*.txt, 200-800 output tokens each) through 128 slots per GPU with continuous
batching; with N GPUs the stream is submitted N times (distinct utterance ids) and dealt over the ranks with dp.shard_indices
(~499 requests per rank: weak scaling).  A "step" there = one engine step (decode rows of the running utterances + prefill rows of
the admitted ones); K timed steps after W warmup steps, then the stream runs to its end for the e2e / RTF figures.

`--workload c2` (BASELINE.json configs[1], the setting of the reference's only published number, README.md:313-325): English vocabulary
704, ONE utterance (the 20-word sentence of C1, T = 108), max_model_len = 400 -> G = 292 tokens; a step = one decode step of the
utterance (2 CFG rows); default 200 timed steps centred on decode step 146.

Every line carries step_ms_p50 / p90 / p99: the distribution of the timed steps' own durations (t3_step_times: from the completion of
the step before to the step's completion, on the host clock that waits on the step's event).

  python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL.  Launched without WORLD_SIZE in the environment, this process starts the N ranks itself
(`python -m torch.distributed.run ... bench.py`, before it touches the GPU) and relays rank 0's JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable copy rate)
KV_BYTES_TOK_STREAM_LAYER = 2 * 16 * 64 * 2
S3_TOKEN_RATE = 25.0           # speech tokens per second of audio (reference s3tokenizer.py:18)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps; 0 = 800 (c3, c4) / 200 (c2)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=("c2", "c3", "c4"), default="c3")
    ap.add_argument("--batch", type=int, default=0, help="utterances (c3) / slots (c4) per GPU; 0 = 32 / 128 (c2: always 1)")
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--max-model-len", type=int, default=0, help="0 = 1000 (c3, c4) / 400 (c2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    a = ap.parse_args()
    if a.workload == "c2":
        a.batch = 1
    if a.batch <= 0:
        a.batch = 32 if a.workload == "c3" else 128
    if a.max_model_len <= 0:
        a.max_model_len = 400 if a.workload == "c2" else 1000
    if a.steps <= 0:
        a.steps = 200 if a.workload == "c2" else 800
    a.vocab = 704 if a.workload == "c2" else 2454
    return a


T_EN, T_ES = 116, 141             # prompt lengths of the two C3 utterances
T_C2 = 108                        # C1 / C2: the 20-word English sentence under the English tokenizer


def longest_prompt(args):
    return T_C2 if getattr(args, "workload", "c3") == "c2" else T_ES


def plan_window(args):
    """(fast-forward steps, first timed decode step, last timed decode step + 1): the timed window is centred on the run midpoint."""
    g_min = args.max_model_len - longest_prompt(args) - 3  # decode steps every utterance of the batch can take
    if args.steps + args.warmup > g_min:
        sys.exit(f"steps + warmup must be <= {g_min} for this workload")
    centre = (args.max_model_len - longest_prompt(args)) // 2
    ff = max(0, min(centre - args.steps // 2 - args.warmup, g_min - args.steps - args.warmup))
    first = 1 + ff + args.warmup                         # the prefill step samples token 0
    return ff, first, first + args.steps


def workload_string(args, first, last):
    if getattr(args, "workload", "c3") == "c2":
        tag = "C2" if (args.max_model_len, args.layers) == (400, 30) else "custom (NOT a BASELINE.json config)"
        return (f"{tag}: t3-model (English, {args.layers}-layer Llama_520M, vocab 704), batch 1 (the 20-word sentence of C1, T={T_C2}), "
                f"max_model_len={args.max_model_len} -> G={args.max_model_len - T_C2} tokens, CFG dual stream (2 rows/step), temperature 0.8 / top-p 0.8 / "
                f"repetition penalty 2.0, stop id masked; timed window = decode steps [{first}, {last}) of the utterance")
    b_en = args.batch // 2
    is_c3 = (args.batch, args.max_model_len, args.layers) == (32, 1000, 30)
    tag = "C3" if is_c3 else "custom (NOT a BASELINE.json config)"
    return (f"{tag}: t3-model-multilingual ({args.layers}-layer Llama_520M, vocab 2454), {b_en} en (T={T_EN}) + {args.batch - b_en} es (T={T_ES}) "
            f"utterances per GPU, max_model_len={args.max_model_len}, CFG dual stream ({2 * args.batch} rows/step), temperature 0.8 / top-p 0.8 / "
            f"repetition penalty 2.0, stop id masked; timed window = decode steps [{first}, {last}) of every utterance")


def check_gpu_count(world):
    """RCCL cannot build a communicator over duplicate devices: say so here, in the bench's own words and before any GPU call, rather than
    from inside init_process_group (torch.cuda.device_count() does not initialise the GPU on this image)."""
    if world <= 1 or os.environ.get("T3_BENCH_BACKEND", "nccl") != "nccl":
        return
    import torch
    ndev = torch.cuda.device_count()
    if world > ndev:
        sys.exit(f"bench.py --gpus {world}: this node exposes {ndev} GPU(s) and the nccl (RCCL) backend needs one GPU per rank; "
                 f"run with --gpus <= {max(ndev, 1)}, or rehearse {world} ranks on the GPUs present with T3_BENCH_BACKEND=gloo")


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as children (nothing in this process has touched the GPU) and relay rank 0."""
    check_gpu_count(args.gpus)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    sys.exit(p.returncode if p.returncode else (0 if line else 1))


def build_requests(E, args, rank):
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    tok = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer.json")))
    p_en, p_es = assemble_prompt_ids(tok["en_mtl_ids"]), assemble_prompt_ids(tok["es_mtl_ids"])
    p_c2 = assemble_prompt_ids(tok["en_english_ids"])
    assert (len(p_en), len(p_es), len(p_c2)) == (T_EN, T_ES, T_C2)
    reqs = []
    for i in range(args.batch):
        prompt = p_c2 if args.workload == "c2" else (p_en if i < args.batch // 2 else p_es)
        uid = rank * args.batch + i                       # global utterance id: shards are disjoint
        sp = E.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=uid,
                             max_tokens=args.max_model_len - len(prompt), ignore_eos=True)
        reqs.append((i, prompt, sp))
    return reqs


def build_c4_requests(E, args, rank, world):
    """The committed C4 stream, submitted `world` times with distinct utterance ids and dealt over the ranks by cost (dp.shard_indices)."""
    from chatterbox_vllm2_amd.dp import shard_indices
    from chatterbox_vllm2_amd.prompt import assemble_prompt_ids
    c4 = json.load(open(os.path.join(ROOT, "tests", "golden", "c4_requests.json")))
    base = c4["requests"]
    n = len(base) * world
    costs = [float(34 + len(base[g % len(base)]["text_ids"]) + 1 + base[g % len(base)]["max_tokens"]) for g in range(n)]
    mine = shard_indices(costs, world)[rank]
    reqs = []
    for g in mine:
        r = base[g % len(base)]
        sp = E.make_sampling(max_tokens=r["max_tokens"], ignore_eos=True, uid=g, **c4["sampling"])
        reqs.append((g, assemble_prompt_ids(r["text_ids"]), sp))
    return reqs, n, c4


def step_percentiles(ms, prefix="step_ms"):
    """p50 / p90 / p99 of the timed steps' own durations (SURVEY.md 8(d) "step-latency histogram")."""
    import numpy as np
    if len(ms) == 0:
        return {}
    a = np.asarray(ms, dtype=np.float64)
    return {f"{prefix}_p50": round(float(np.percentile(a, 50)), 4), f"{prefix}_p90": round(float(np.percentile(a, 90)), 4),
            f"{prefix}_p99": round(float(np.percentile(a, 99)), 4), f"{prefix}_max": round(float(a.max()), 4)}


def file_sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def rtf_stats(eng, reqs, n_tokens):
    """Per-request RTF from the engine's own clock marks: (finished - admitted) / (tokens / 25); also from add_request (queue wait included)."""
    import numpy as np
    run, tot = [], []
    for rid, _, _ in reqs:
        t_add, t_adm, _, t_fin = eng.timing(rid)
        audio_s = n_tokens[rid] / S3_TOKEN_RATE
        run.append((t_fin - t_adm) / audio_s); tot.append((t_fin - t_add) / audio_s)
    run, tot = np.array(run), np.array(tot)
    return {"rtf_p50": round(float(np.percentile(run, 50)), 5), "rtf_p90": round(float(np.percentile(run, 90)), 5), "rtf_max": round(float(run.max()), 5),
            "rtf_incl_queue_p50": round(float(np.percentile(tot, 50)), 5), "rtf_incl_queue_p90": round(float(np.percentile(tot, 90)), 5),
            "rtf_definition": "T3 wall time of request i from admission to its last token / (n_tokens_i / 25 tokens per second of audio); incl_queue: from add_request"}


def run_pass(eng, reqs, cond, ff, warmup, steps, sync, profile=False, only=None):
    """prefill (untimed) -> fast-forward + warmup decode steps -> barrier -> K timed decode steps -> barrier.  Returns dict."""
    import torch
    for rid, prompt, sp in reqs:
        eng.add_request(rid, prompt, cond, sp)
    t0 = time.perf_counter()
    n_pref = 0
    while True:                                          # prefill steps (all prompts) until every utterance decodes
        r = eng.step(); n_pref += 1
        if r.n_prefill_rows == 0 or r.n_waiting == 0 and r.n_sampled == len(reqs):
            break
    torch.cuda.synchronize()
    prefill_s = time.perf_counter() - t0
    assert eng.run_steps(ff + warmup) == ff + warmup
    eng.set_profile(profile, only)
    eng.reset_stats()
    sync()
    t0 = time.perf_counter()
    done = eng.run_steps(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sync()
    eng.set_profile(False)
    assert done == steps, f"only {done} of {steps} steps ran (fast-forward + warmup + steps must stay below max_model_len - longest prompt)"
    st = eng.stats()
    assert st.decode_steps == steps and st.tokens_generated == steps * len(reqs)
    step_ms, _ = eng.step_times(steps)
    kern = {k: eng.kernel_ms(k) for k in __import__("chatterbox_vllm2_amd.engine", fromlist=["x"]).KERNEL_CLASSES} if profile else {}
    # drain: finish the utterances quickly (not timed) so the engine is reusable
    eng.run_until_done()
    for rid, _, _ in reqs:
        eng.release(rid)
    return dict(dt=dt, prefill_s=prefill_s, prefill_steps=n_pref, stats=st, kern=kern, step_ms=step_ms)


def run_e2e(eng, reqs, cond):
    """Whole utterances, nothing skipped: add_request -> prefill -> every decode step to the length limit; tokens / wall + per-request RTF."""
    import torch
    eng.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rid, prompt, sp in reqs:
        eng.add_request(rid, prompt, cond, sp)
    eng.run_until_done()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.stats()
    n_tokens = {rid: len(eng.get_output(rid)[0]) for rid, _, _ in reqs}
    out = {"value": round(st.tokens_generated / dt, 2), "unit": "speech-tokens/s", "tokens": int(st.tokens_generated), "wall_s": round(dt, 4),
           "engine_steps": int(st.steps), "includes": "add_request (host buffers in), prefill, every decode step, token read-back",
           "audio_seconds_per_second": round(st.tokens_generated / S3_TOKEN_RATE / dt, 1)}
    out.update(rtf_stats(eng, reqs, n_tokens))
    for rid, _, _ in reqs:
        eng.release(rid)
    return out


def run_c4(eng, reqs, cond, warmup, steps, sync):
    """Continuous batching: all requests queued, W untimed steps, barrier, EXACTLY K timed engine steps, barrier, then the rest of the
    stream (untimed for `value`, but inside the e2e wall clock)."""
    import torch
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for rid, prompt, sp in reqs:
        eng.add_request(rid, prompt, cond, sp)
    assert eng.run_steps(warmup) == warmup
    eng.reset_stats()
    sync()
    t0 = time.perf_counter()
    done = eng.run_steps(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sync()
    st = eng.stats()
    if done != steps:
        sys.exit(f"only {done} of {steps} engine steps were left after {warmup} warmup steps: lower --steps for this request stream")
    tokens_window = int(st.tokens_generated)
    step_ms, step_rows = eng.step_times(steps)
    window = dict(step_ms=step_ms, step_rows=step_rows, dt=dt, tokens=tokens_window, decode_only_steps=int(st.decode_steps), prefill_rows=int(st.prefill_rows), decode_rows=int(st.decode_rows),
                  ms_decode_only=st.gpu_ms_decode / max(1, st.decode_steps), algo_bytes_decode=st.algo_bytes_decode, gpu_ms_decode=st.gpu_ms_decode,
                  mean_ctx=st.sum_ctx_decode / max(1, st.decode_rows))
    eng.run_until_done()
    torch.cuda.synchronize()
    dt_all = time.perf_counter() - t_all
    n_tokens = {rid: len(eng.get_output(rid)[0]) for rid, _, _ in reqs}
    total = sum(n_tokens.values())
    e2e = {"value": round(total / dt_all, 2), "unit": "speech-tokens/s (this rank)", "tokens": total, "wall_s": round(dt_all, 4), "requests": len(reqs),
           "audio_seconds_per_second": round(total / S3_TOKEN_RATE / dt_all, 1),
           "includes": "admission, chunked prefill interleaved with decode, every decode step, retirement"}
    e2e.update(rtf_stats(eng, reqs, n_tokens))
    st2 = eng.stats()
    assert st2.kv_blocks_free == st2.kv_blocks_total
    for rid, _, _ in reqs:
        eng.release(rid)
    return window, e2e


def cpu_baseline(weights, args):
    """The oracle (a port, kind="port") timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    ncores = int(os.environ.get("T3_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))   # the box grants a 16-core share
    O.set_threads(ncores)
    B, ctx = min(args.batch, 32), (500 if args.workload != "c2" else 250)
    steps = 5 if B > 1 else 40                          # ~13 s of CPU work on the box's 16-core share (B = 1: ~0.3 s per step)
    m = O.OracleModel(args.layers, args.vocab, max_pos=ctx + steps + 2, n_streams=2 * B).load(weights)
    m.decode_steps_timing(1, ctx, 1)                     # touch the weights once
    t0 = time.perf_counter()
    m.decode_steps_timing(B, ctx, steps)
    dt = time.perf_counter() - t0
    m.close()
    return {"value": round(B * steps / dt, 3), "unit": "speech-tokens/s", "cores": ncores, "kind": "port",
            "sample": f"C oracle (OpenMP, {ncores} threads): {steps} decode steps of a B={B} batch ({2 * B} CFG rows, {args.layers} layers) "
                      f"at context {ctx}, weights resident in RAM; {dt:.1f} s of CPU work"}


def dominant_kernel_roofline(prof, dom_only, args):
    kern = prof["kern"]; pst = prof["stats"]
    tot = {k: v[0] * v[1] for k, v in kern.items()}
    dom = max(tot, key=tot.get)
    wbytes = {"gemm_qkv": 3072 * 1024 * 2, "gemm_o": 1024 * 1024 * 2, "gemm_gateup": 8192 * 1024 * 2, "gemm_down": 4096 * 1024 * 2,
              "gemm_head": 8194 * 1024 * 2,
              "sampler": args.batch * 3 * 8208 * 2, "embed": args.batch * 2 * 3 * 2048}      # per launch: two logit rows + the counts per utterance; table rows in, one row out
    if dom == "attention":      # KV read of one layer: sum over the 2B rows of ctx * 4096 B (SURVEY 8d per-unit figure / 30 layers)
        algo = pst.sum_ctx_decode / pst.decode_steps * KV_BYTES_TOK_STREAM_LAYER
    else:
        algo = wbytes.get(dom, 0)
    ms = dom_only["kern"][dom][0]                  # events around this class only
    # HBM bytes per launch: NOT measured by this run (PMC counters need rocprofv3 around the process).  profiles/traffic.json holds the
    # bytes-fetched / algorithmic-bytes ratio of the committed FETCH_SIZE pass (gfx950 x2 correction) together with the sha256 of the
    # kernel source it was collected on; the ratio is applied to this run's algorithmic bytes ONLY while that source is unchanged,
    # otherwise traffic is null (a stale ratio is not a measurement of this kernel).
    traffic, traffic_source = None, "none: profiles/traffic.json has no entry for this kernel class"
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        tj = json.load(open(tf))
        rec = tj.get(dom)
        if rec and "ratio" in rec:
            src = os.path.join(ROOT, "chatterbox-vllm2_amd", "csrc", rec.get("source_file", "t3_attention.hip" if dom == "attention" else "t3_gemm.hip"))
            now, then = file_sha16(src), rec.get("source_sha16")
            if then == now:
                traffic = round(rec["ratio"] * algo)
                traffic_source = (f"ratio {rec['ratio']:.4f} of the committed rocprofv3 --pmc FETCH_SIZE pass {rec.get('pass', 'profiles/traffic.json')} "
                                  f"(x2 gfx950 correction) applied to this run's algorithmic bytes; kernel source unchanged since (sha256 {now})")
            else:
                traffic_source = f"null: {os.path.basename(src)} changed since the committed FETCH_SIZE pass (sha256 {then} -> {now}); re-run tools/run_profiles.sh"
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "algo_bytes_per_launch": round(algo), "avg_launch_ms": round(ms, 5), "launches": kern[dom][1],
            "timing": "HIP events on the engine's stream, as the start / stop events of every launch of this kernel class (hipExtLaunchKernelGGL: "
                      "the dispatch's begin / end) in a pass of the same steps; the rocprofv3 average of the same kernel is in profiles/README.md"}
    return roof, {k: round(tot[k] / pst.decode_steps, 4) for k in tot}


def main():
    args = parse()
    ff, first, last = plan_window(args) if args.workload != "c4" else (0, 0, 0)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    check_gpu_count(world)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the T3 engine has no CPU path")
    # T3_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box); default RCCL.
    # T3_BENCH_FORCE_DIST=1 creates the process group at N = 1 as well (RCCL init + barrier + all-reduce rehearsed on one GPU).
    backend = os.environ.get("T3_BENCH_BACKEND", "nccl")
    use_dist = world > 1 or os.environ.get("T3_BENCH_FORCE_DIST", "0") == "1"
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    from chatterbox_vllm2_amd import engine as E
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors

    weights = list(synthetic_tensors(args.layers, args.vocab, 1234))
    cond = synthetic_cond_emb(1)
    share = max(1, (world if backend != "nccl" else 1))                     # ranks sharing one GPU in a gloo rehearsal
    eng = E.T3Engine(n_layers=args.layers, text_vocab=args.vocab, max_model_len=args.max_model_len, max_seqs=args.batch,
                     device_id=dev, gpu_memory_utilization=(0.6 if args.workload == "c4" else 0.5) / share, max_batched_rows=8192,
                     enforce_eager=bool(int(os.environ.get("T3_EAGER", "0"))))
    eng.load_tensors(weights); eng.finalize()
    common = {"metric": "speech-tokens/sec/GPU (T3 decode, batch=32) + p50 RTF", "unit": "speech-tokens/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic"}

    if args.workload == "c4":
        reqs, n_global, c4 = build_c4_requests(E, args, rank, world)
        window, e2e = run_c4(eng, reqs, cond, args.warmup, args.steps, sync)
        dt = max_over_ranks(window["dt"])
        tokens = sum_over_ranks(window["tokens"])
        e2e_tokens = sum_over_ranks(e2e["tokens"]); e2e_wall = max_over_ranks(e2e["wall_s"])
        if use_dist:
            dist.barrier()
        eng.close()
        if rank == 0:
            out = dict(common)
            out.update({"value": round(tokens / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 4),
                        "config": {"workload": f"C4{' x ' + str(world) + ' GPUs (C5)' if world > 1 else ''}: {n_global} requests (tests/golden/c4_requests.json"
                                               f"{' submitted ' + str(world) + ' times with distinct utterance ids' if world > 1 else ''}: sentences of the reference's "
                                               f"docs/benchmark-text-1/2/fr-1/zh-1.txt, 13-387 text ids, 200-800 output tokens each), dealt over the ranks by cost, "
                                               f"{args.batch} slots per GPU, continuous batching, {args.layers} layers, max_model_len {args.max_model_len}; a step = one "
                                               f"engine step (decode rows + prefill rows of newly admitted requests); timed window = engine steps [{args.warmup}, {args.warmup + args.steps})",
                                   "batch_per_gpu": args.batch, "global_batch": world * args.batch, "requests_rank0": len(reqs), "max_model_len": args.max_model_len,
                                   "layers": args.layers, "parallelism": f"dp{world} (utterance shards, no collective inside a step)",
                                   "weights": "seeded synthetic N(0,0.02^2), seed 1234"},
                        "tokens_per_s_per_gpu": round(tokens / dt / world, 2),
                        "rtf_p50": e2e["rtf_p50"], "rtf_p90": e2e["rtf_p90"],
                        **step_percentiles(window["step_ms"]),
                        "step_ms_by_kind_rank0": {"decode_only": dict(n=int((window["step_rows"] > 0).sum()), **step_percentiles(window["step_ms"][window["step_rows"] > 0], "ms")),
                                                  "with_prefill_rows": dict(n=int((window["step_rows"] < 0).sum()), mean_rows=round(float(-window["step_rows"][window["step_rows"] < 0].mean()), 1) if (window["step_rows"] < 0).any() else 0,
                                                                            **step_percentiles(window["step_ms"][window["step_rows"] < 0], "ms"))},
                        "window_rank0": {"decode_only_steps": window["decode_only_steps"], "prefill_rows": window["prefill_rows"], "decode_rows": window["decode_rows"],
                                         "ms_per_decode_only_step": round(window["ms_decode_only"], 4), "mean_ctx_decode": round(window["mean_ctx"], 1),
                                         "step_hbm_frac_decode_only": round(window["algo_bytes_decode"] / max(1e-9, window["gpu_ms_decode"] * 1e-3) / 8e12, 4)},
                        "e2e": dict(e2e, value_all_ranks=round(e2e_tokens / e2e_wall, 2), wall_s_max_over_ranks=round(e2e_wall, 4))})
            print(json.dumps(out), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    reqs = build_requests(E, args, rank)
    res = run_pass(eng, reqs, cond, ff, args.warmup, args.steps, sync, profile=False)
    dt = max_over_ranks(res["dt"])
    st = res["stats"]

    prof = dom_only = e2e = None
    if rank == 0 and not args.no_profile_pass:
        # pass 2: HIP events around every kernel class (eager launches) -> which class dominates, and the per-class table;
        # pass 3: events around the dominant class only, the rest of the step undisturbed -> its average launch duration
        nsync = lambda: torch.cuda.synchronize()
        prof = run_pass(eng, reqs, cond, ff, args.warmup, args.steps, nsync, profile=True)
        tot = {k: v[0] * v[1] for k, v in prof["kern"].items()}
        dom_only = run_pass(eng, reqs, cond, ff, args.warmup, args.steps, nsync, profile=True, only=max(tot, key=tot.get))
    if rank == 0 and not args.no_e2e:
        e2e = run_e2e(eng, reqs, cond)
    if use_dist:
        dist.barrier()
    eng.close()

    if rank == 0:
        total_tokens = world * args.batch * args.steps
        value = total_tokens / dt
        bytes_step = st.algo_bytes_decode / st.decode_steps
        mean_T = T_C2 if args.workload == "c2" else (T_EN * (args.batch // 2) + T_ES * (args.batch - args.batch // 2)) / args.batch
        out = dict(common)
        out.update({
            "value": round(value, 2), "ms_per_step": round(dt / args.steps * 1e3, 4),
            "config": {"workload": workload_string(args, first, last),
                       "ctx_first": round(mean_T + first, 1), "ctx_last": round(mean_T + last - 1, 1),
                       "fast_forward_steps": ff, "batch_per_gpu": args.batch, "global_batch": world * args.batch, "max_model_len": args.max_model_len,
                       "layers": args.layers, "parallelism": f"dp{world} (utterance shards, no collective inside a step)",
                       "weights": "seeded synthetic N(0,0.02^2), seed 1234"},
            "tokens_per_s_per_gpu": round(value / world, 2),
            # in the timed window every utterance emits one token per step: its RTF there is the step time x 25.  The per-request figure
            # over whole utterances (prefill included) is e2e.rtf_p50 / rtf_p90.
            "rtf_p50": round((dt / args.steps) * S3_TOKEN_RATE, 5),
            **step_percentiles(res["step_ms"]),
            "prefill_ms": round(res["prefill_s"] * 1e3, 2), "prefill_steps": res["prefill_steps"],
            "mean_ctx": round(st.sum_ctx_decode / st.decode_rows, 1),
            "step_roofline": {"bound": "hbm", "algo_bytes_per_step": round(bytes_step), "achieved": round(bytes_step / (dt / args.steps) / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
        })
        if e2e:
            out["e2e"] = e2e
        if prof:
            out["roofline"], out["kernel_ms_per_step_all_classes_evented"] = dominant_kernel_roofline(prof, dom_only, args)   # eager + 2 events per kernel: over-reports
            out["profiled_ms_per_step"] = round(prof["dt"] / args.steps * 1e3, 4)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(weights, args)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
