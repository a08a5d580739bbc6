#!/usr/bin/env python3
"""bench.py -- T3 decode throughput on MI355X (BASELINE.json metric: speech-tokens/s/GPU, batch 32, + p50 RTF).

Workload (config.workload = "C3"): multilingual vocab 2454, B = 32 utterances per GPU = 16 en prompts
(T = 116) + 16 es prompts (T = 141), max_model_len = 1000, real layer count (30), bf16 weights, seeded
synthetic weights (no checkpoint offline), the reference's sampling defaults (temperature 0.8, top-p 0.8,
repetition penalty 2.0, tts.py:377,416) with the stop id masked (fixed-length generation).
A "step" = one pass of the hot path over the batch = one decode step of all 32 utterances (64 CFG rows).
Prefill, fast-forward and warmup steps are outside the timed region; inputs (weights, prompts, KV) are resident in HBM.

The K timed steps are CENTRED on the middle of the utterances' lives (decode step (max_model_len - longest prompt) / 2, where the
context equals the whole-run mean, ~560 tokens for C3) whatever K is: the engine is fast-forwarded with untimed decode steps
first, so `--steps 20` and `--steps 800` measure the same operating point (config.ctx_first / ctx_last say which).

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=800)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--max-model-len", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    return ap.parse_args()


T_EN, T_ES = 116, 141             # prompt lengths of the two C3 utterances


def plan_window(args):
    """(fast-forward steps, first timed decode step, last timed decode step + 1): the timed window is centred on the run midpoint."""
    g_min = args.max_model_len - T_ES - 3                 # decode steps every utterance of the batch can take
    if args.steps + args.warmup > g_min:
        sys.exit(f"steps + warmup must be <= {g_min} for this workload")
    centre = (args.max_model_len - T_ES) // 2
    ff = max(0, min(centre - args.steps // 2 - args.warmup, g_min - args.steps - args.warmup))
    first = 1 + ff + args.warmup                         # the prefill step samples token 0
    return ff, first, first + args.steps


def workload_string(args, first, last):
    b_en = args.batch // 2
    is_c3 = (args.batch, args.max_model_len, args.layers) == (32, 1000, 30)
    tag = "C3" if is_c3 else "custom (NOT a BASELINE.json config)"
    return (f"{tag}: t3-model-multilingual ({args.layers}-layer Llama_520M, vocab 2454), {b_en} en (T={T_EN}) + {args.batch - b_en} es (T={T_ES}) "
            f"utterances per GPU, max_model_len={args.max_model_len}, CFG dual stream ({2 * args.batch} rows/step), temperature 0.8 / top-p 0.8 / "
            f"repetition penalty 2.0, stop id masked; timed window = decode steps [{first}, {last}) of every utterance")


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as children (nothing in this process has touched the GPU) and relay rank 0."""
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
    reqs = []
    for i in range(args.batch):
        prompt = p_en if i < args.batch // 2 else p_es
        uid = rank * args.batch + i                       # global utterance id: shards are disjoint
        sp = E.make_sampling(temperature=0.8, top_p=0.8, repetition_penalty=2.0, seed=0, uid=uid,
                             max_tokens=args.max_model_len - len(prompt), ignore_eos=True)
        reqs.append((i, prompt, sp))
    return reqs


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
    kern = {k: eng.kernel_ms(k) for k in __import__("chatterbox_vllm2_amd.engine", fromlist=["x"]).KERNEL_CLASSES} if profile else {}
    # drain: finish the utterances quickly (not timed) so the engine is reusable
    eng.run_until_done()
    for rid, _, _ in reqs:
        eng.release(rid)
    return dict(dt=dt, prefill_s=prefill_s, prefill_steps=n_pref, stats=st, kern=kern)


def cpu_baseline(weights, args):
    """The oracle (a port, kind="port") timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    ncores = int(os.environ.get("T3_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))   # the box grants a 16-core share
    O.set_threads(ncores)
    B, ctx, steps = args.batch, 500, 5                   # ~13 s of CPU work on the box's 16-core share
    m = O.OracleModel(args.layers, 2454, max_pos=ctx + steps + 2, n_streams=2 * B).load(weights)
    m.decode_steps_timing(1, ctx, 1)                     # touch the weights once
    t0 = time.perf_counter()
    m.decode_steps_timing(B, ctx, steps)
    dt = time.perf_counter() - t0
    m.close()
    return {"value": round(B * steps / dt, 3), "unit": "speech-tokens/s", "cores": ncores, "kind": "port",
            "sample": f"C oracle (OpenMP, {ncores} threads): {steps} decode steps of the same B={B} batch ({2 * B} CFG rows, {args.layers} layers) "
                      f"at context {ctx}, weights resident in RAM; {dt:.1f} s of CPU work"}


def main():
    args = parse()
    ff, first, last = plan_window(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the T3 engine has no CPU path")
    # T3_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box); default RCCL
    backend = os.environ.get("T3_BENCH_BACKEND", "nccl")
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from chatterbox_vllm2_amd import engine as E
    from chatterbox_vllm2_amd.weights import synthetic_cond_emb, synthetic_tensors

    weights = list(synthetic_tensors(args.layers, 2454, 1234))
    cond = synthetic_cond_emb(1)
    eng = E.T3Engine(n_layers=args.layers, text_vocab=2454, max_model_len=args.max_model_len, max_seqs=args.batch,
                     device_id=dev, gpu_memory_utilization=0.5 / max(1, (world if backend != 'nccl' else 1)), max_batched_rows=8192, enforce_eager=bool(int(os.environ.get('T3_EAGER', '0'))))
    eng.load_tensors(weights); eng.finalize()
    reqs = build_requests(E, args, rank)

    res = run_pass(eng, reqs, cond, ff, args.warmup, args.steps, sync, profile=False)
    t = torch.tensor([res["dt"]], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    st = res["stats"]

    prof = dom_only = None
    if rank == 0 and not args.no_profile_pass:
        # pass 2: HIP events around every kernel class (eager launches) -> which class dominates, and the per-class table;
        # pass 3: events around the dominant class only, the rest of the step undisturbed -> its average launch duration
        nsync = lambda: torch.cuda.synchronize()
        prof = run_pass(eng, reqs, cond, ff, args.warmup, args.steps, nsync, profile=True)
        tot = {k: v[0] * v[1] for k, v in prof["kern"].items()}
        dom_only = run_pass(eng, reqs, cond, ff, args.warmup, args.steps, nsync, profile=True, only=max(tot, key=tot.get))
    if world > 1:
        dist.barrier()
    eng.close()

    if rank == 0:
        total_tokens = world * args.batch * args.steps
        value = total_tokens / dt
        bytes_step = st.algo_bytes_decode / st.decode_steps
        out = {
            "metric": "speech-tokens/sec/GPU (T3 decode, batch=32) + p50 RTF", "value": round(value, 2), "unit": "speech-tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload_string(args, first, last),
                       "ctx_first": round((T_EN * (args.batch // 2) + T_ES * (args.batch - args.batch // 2)) / args.batch + first, 1),
                       "ctx_last": round((T_EN * (args.batch // 2) + T_ES * (args.batch - args.batch // 2)) / args.batch + last - 1, 1),
                       "fast_forward_steps": ff, "batch_per_gpu": args.batch, "global_batch": world * args.batch, "max_model_len": args.max_model_len,
                       "layers": args.layers, "parallelism": f"dp{world} (utterance shards, no collective inside a step)",
                       "weights": "seeded synthetic N(0,0.02^2), seed 1234"},
            "tokens_per_s_per_gpu": round(value / world, 2),
            "rtf_p50": round((dt / args.steps) * 25.0, 5),      # every utterance emits one token per step: wall / (tokens/25)
            "prefill_ms": round(res["prefill_s"] * 1e3, 2), "prefill_steps": res["prefill_steps"],
            "mean_ctx": round(st.sum_ctx_decode / st.decode_rows, 1),
            "step_roofline": {"bound": "hbm", "algo_bytes_per_step": round(bytes_step), "achieved": round(bytes_step / (dt / args.steps) / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if prof:
            kern = prof["kern"]; pst = prof["stats"]
            tot = {k: v[0] * v[1] for k, v in kern.items()}
            dom = max(tot, key=tot.get)
            wbytes = {"gemm_qkv": 3072 * 1024 * 2, "gemm_o": 1024 * 1024 * 2, "gemm_gateup": 8192 * 1024 * 2, "gemm_down": 4096 * 1024 * 2,
                      "gemm_head": 8194 * 1024 * 2}
            if dom == "attention":      # KV read of one layer: sum over the 2B rows of ctx * 4096 B (SURVEY 8d per-unit figure / 30 layers)
                algo = pst.sum_ctx_decode / pst.decode_steps * KV_BYTES_TOK_STREAM_LAYER
            else:
                algo = wbytes.get(dom, 0)
            ms = dom_only["kern"][dom][0]                  # events around this class only
            # HBM bytes per launch from the PMC pass recorded in profiles/traffic.json (FETCH_SIZE, gfx950 x2 correction):
            # the measured bytes/algorithmic-bytes ratio of that pass applied to this run's algorithmic bytes per launch
            traffic = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf):
                rec = json.load(open(tf)).get(dom)
                if rec and "ratio" in rec:
                    traffic = round(rec["ratio"] * algo)
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "algo_bytes_per_launch": round(algo), "avg_launch_ms": round(ms, 5), "launches": kern[dom][1]}
            out["roofline"]["timing"] = "HIP events on the engine's stream around every launch of this kernel class in a pass of the same steps"
            # the committed rocprofv3 --kernel-trace --stats average of the same kernel on the same workload, for comparison (an event
            # pair around a launch costs ~2-3 us on top of the kernel: DESIGN.md section 6)
            ks = os.path.join(ROOT, "profiles", "r02_d_kernel_stats_c3.csv")
            rp_name = {"attention": "attention_kernel<4, true, true>"}.get(dom)
            if rp_name and os.path.exists(ks) and args.batch == 32 and args.layers == 30:
                import csv
                for row in csv.DictReader(open(ks)):
                    if rp_name in row["Name"]:
                        rp_ms = float(row["AverageNs"]) * 1e-6
                        out["roofline"]["rocprof_avg_launch_ms"] = round(rp_ms, 5)
                        out["roofline"]["rocprof_frac"] = round(algo / (rp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                        out["roofline"]["rocprof_source"] = "profiles/r02_d_kernel_stats_c3.csv (all 64-row launches of a 100-step run of this workload)"
                        break
            out["kernel_ms_per_step_all_classes_evented"] = {k: round(tot[k] / pst.decode_steps, 4) for k in tot}   # eager + 2 events per kernel: over-reports
            out["profiled_ms_per_step"] = round(prof["dt"] / args.steps * 1e3, 4)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(weights, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
