"""profiles/traffic.json from a FETCH_SIZE summary of tools/run_profiles.sh (tools/pmc_summarize.py output).

usage: python tools/update_traffic.py profiles/<tag>_pmc_fetch_size_by_kernel.json

For every kernel class bench.py can name as dominant: HBM bytes per launch = 2 * 1024 * FETCH_SIZE (KiB; gfx950 counts half of a wide
coalesced stream, MI355X_MICROARCH.md "HBM"), the algorithmic bytes of the same launches, their ratio, and the sha256 of the kernel source
the pass ran on -- bench.py applies the ratio only while that source is unchanged (otherwise roofline.traffic is null).
The pass is `bench.py --steps 100 --warmup 5` on C3: its 64-row decode launches cover every decode step of the utterances' lives
(fast-forward + window + drain, twice with the e2e run), mean context 558.5 -> 558.5 x 64 rows x 4096 B algorithmic per attention launch."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chatterbox-vllm2_amd", "csrc")
src = sys.argv[1]
k = json.load(open(src))["kernels"]
sha = lambda f: hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:16]


def pick(pred):
    best = None
    for name, v in k.items():
        if pred(name) and (best is None or v["launches"] > best[1]["launches"]):
            best = (name, v)
    return best


classes = {   # class -> (predicate on the kernel name of its 64-row decode form, source file, algorithmic bytes per launch)
    "attention": (lambda n: n.startswith("attention_kernel<4, true, true>"), "t3_attention.hip", 558.5 * 64 * 4096),
    "gemm_gateup": (lambda n: "gemm2_split_kernel<2>" in n or n.startswith("gemm2_kernel<2, 4, 3,"), "t3_gemm.hip", 8192 * 1024 * 2 + 64 * 1024 * 2 + 64 * 4096 * 2),
    "gemm_qkv": (lambda n: n.startswith("gemm2_kernel<1, 3, 1, 4, 8, true, 8"), "t3_gemm.hip", 3072 * 1024 * 2 + 64 * 1024 * 2 + 64 * 3072 * 2),
    "gemm_down": (lambda n: n.startswith("gemm2_kernel<1, 1, 2, 16, 8, false, 8"), "t3_gemm.hip", 4096 * 1024 * 2 + 64 * 4096 * 2 + 2 * 64 * 1024 * 2),
    "gemm_o": (lambda n: n.startswith("gemm2_kernel<1, 1, 2, 16, 2, false, 2"), "t3_gemm.hip", 1024 * 1024 * 2 + 3 * 64 * 1024 * 2),
}
out = {"_how": "rocprofv3 --pmc FETCH_SIZE --kernel-trace (a pass of its own) -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile-pass "
               "(tools/run_profiles.sh), summarised on the box by tools/pmc_summarize.py, turned into this file by tools/update_traffic.py.  FETCH_SIZE is in KiB "
               "and counts HALF of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM) -> bytes = 2 * 1024 * FETCH_SIZE; Infinity-Cache hits are counted too.",
       "_pass": os.path.relpath(src, ROOT)}
for cls, (pred, f, algo) in classes.items():
    hit = pick(pred)
    if not hit:
        continue
    name, v = hit
    hbm = 2 * 1024 * v["avg"]
    out[cls] = {"kernel": name, "launches": v["launches"], "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algo, "ratio": hbm / algo,
                "source_file": f, "source_sha16": sha(f), "pass": os.path.relpath(src, ROOT)}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({c: round(v["ratio"], 4) for c, v in out.items() if isinstance(v, dict)}))
