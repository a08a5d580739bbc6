"""The C-ABI library loads and exports every symbol include/t3_engine.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "t3_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(t3k?_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_documented_surface():
    fns = header_functions()
    for must in ("t3_create", "t3_load_tensor", "t3_finalize_weights", "t3_add_request", "t3_step", "t3_run_until_done",
                 "t3_get_output", "t3_debug_logits", "t3_stats", "t3_destroy", "t3k_gemm", "t3k_rope_attention", "t3k_sample"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from chatterbox_vllm2_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    lib = ctypes.CDLL(engine.LIB_PATH)
    missing = [f for f in header_functions() if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(engine.ABI_SYMBOLS) == header_functions()


def test_struct_layouts_match_the_header():
    """sizeof/offsets the C side assumes (T3Sampling is shared verbatim with the oracle's struct)."""
    from chatterbox_vllm2_amd import engine
    from oracle import oracle as O
    assert ctypes.sizeof(engine.T3Sampling) == 64 == ctypes.sizeof(O.Sampling)
    assert engine.T3Sampling.seed.offset == 40 and engine.T3Sampling.uid.offset == 48
    assert ctypes.sizeof(engine.T3EngineConfig) == 56 and engine.T3EngineConfig.kv_bytes.offset == 24
    assert ctypes.sizeof(engine.T3StepResult) == 24 + 64 * 8
    assert ctypes.sizeof(engine.T3Stats) == 13 * 8


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chatterbox_vllm2_amd import engine
    with pytest.raises(engine.T3Error, match="no HIP device"):
        engine.T3Engine(n_layers=2)
    with pytest.raises(engine.T3Error):
        engine.k_expf(torch.zeros(4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "chatterbox-vllm2_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include[^\n]*oracle|libt3oracle|orc_[a-z]+\s*\(", txt, flags=re.M), \
                    f"{f} uses the oracle"
