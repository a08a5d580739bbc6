"""bench.py host logic that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_ranks_than_gpus_is_refused_in_the_bench_own_words():
    """`--gpus N` on a node with fewer GPUs must stop with bench.py's message before any GPU call or rank is started
    (RCCL would fail inside init_process_group over duplicate devices).  This container has no GPU: N = 2 > 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "T3_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        return
    assert p.returncode != 0 and "needs one GPU per rank" in p.stderr and "T3_BENCH_BACKEND=gloo" in p.stderr and "Traceback" not in p.stderr


def test_window_plans():
    sys.path.insert(0, ROOT)
    import bench
    a = type("A", (), dict(workload="c2", max_model_len=400, steps=200, warmup=20, batch=1, layers=30))()
    ff, first, last = bench.plan_window(a)
    assert last - first == 200 and first == 1 + ff + 20 and (first + last) // 2 in range(140, 152)          # centred on decode step 146 of 292
    assert bench.workload_string(a, first, last).startswith("C2:")
    a = type("A", (), dict(workload="c3", max_model_len=1000, steps=20, warmup=5, batch=32, layers=30))()
    ff, first, last = bench.plan_window(a)
    assert bench.workload_string(a, first, last).startswith("C3:") and abs((first + last) // 2 - 430) <= 2
