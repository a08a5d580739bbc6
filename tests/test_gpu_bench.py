"""bench.py as the driver runs it, on the multi-rank paths that a one-GPU box can rehearse: the first 8-GPU run must
not also be the first run of this code.
  * `--gpus 2` self-launching (torch.distributed.run child) with the gloo backend, two engine processes sharing the box's GPU;
  * `--gpus 1` with the process group FORCED on over RCCL (world 1): communicator creation, barrier and all-reduce on hardware;
  * `--workload c4 --gpus 2`: the committed request stream dealt over two ranks (BASELINE.json configs[4] is `--gpus 8 --workload c4`).
Reduced depth (2 layers) and few steps: these check the plumbing and the JSON contract, not the speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + "\n---\n" + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def test_bench_two_ranks_gloo_rehearsal():
    out = _bench(["--gpus", "2", "--steps", "5", "--warmup", "2", "--layers", "2", "--no-cpu-baseline", "--no-profile-pass"], {"T3_BENCH_BACKEND": "gloo"})
    assert CONTRACT_KEYS <= set(out)
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["config"]["global_batch"] == 64 and out["config"]["batch_per_gpu"] == 32 and "dp2" in out["config"]["parallelism"]
    assert out["value"] > 0 and abs(out["value"] - 64 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-2 * out["value"]      # whole-job aggregate
    e = out["e2e"]                                           # rank 0's whole utterances: 16 x 884 + 16 x 859 tokens
    assert e["tokens"] == 16 * 884 + 16 * 859 and 0 < e["rtf_p50"] <= e["rtf_p90"] <= e["rtf_max"]


def test_bench_one_rank_rccl_process_group():
    """RCCL itself (backend nccl) on the one GPU: init_process_group, barrier, all_reduce(MAX) around the timed region."""
    out = _bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--layers", "2", "--no-cpu-baseline", "--no-profile-pass", "--no-e2e"],
                 {"T3_BENCH_FORCE_DIST": "1", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["global_batch"] == 32


def test_bench_c4_two_ranks():
    out = _bench(["--workload", "c4", "--gpus", "2", "--steps", "40", "--warmup", "5", "--layers", "2", "--no-cpu-baseline", "--no-profile-pass"],
                 {"T3_BENCH_BACKEND": "gloo"})
    assert CONTRACT_KEYS <= set(out) and out["n_gpus"] == 2 and out["config"]["global_batch"] == 256
    assert "C4" in out["config"]["workload"] and out["config"]["requests_rank0"] in (498, 499, 500)
    e = out["e2e"]
    assert e["requests"] == out["config"]["requests_rank0"] and 0 < e["rtf_p50"] <= e["rtf_p90"] and e["rtf_incl_queue_p90"] >= e["rtf_p90"]
    assert e["value_all_ranks"] > e["value"] * 1.2           # two ranks' tokens over the slower rank's wall


def test_bench_c2_line_and_step_percentiles():
    """`--workload c2` = BASELINE.json configs[1] (English vocabulary 704, the T = 108 prompt, B = 1, max_model_len 400): the line a reader
    compares with the reference's only published number; every line carries the step-latency percentiles SURVEY.md 8(d) asks for."""
    out = _bench(["--workload", "c2", "--steps", "30", "--warmup", "5", "--layers", "2", "--no-cpu-baseline"], {})
    assert CONTRACT_KEYS <= set(out) and out["config"]["workload"].startswith("custom")       # 2 layers: not the BASELINE config, and labelled so
    assert out["config"]["batch_per_gpu"] == 1 and out["config"]["max_model_len"] == 400 and "T=108" in out["config"]["workload"] and "vocab 704" in out["config"]["workload"]
    assert 0 < out["step_ms_p50"] <= out["step_ms_p90"] <= out["step_ms_p99"] <= out["step_ms_max"]
    assert out["e2e"]["tokens"] == 292 and out["roofline"]["frac"] > 0 and "traffic_source" in out["roofline"]
