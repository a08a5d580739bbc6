"""Build-time guard (no GPU): the GEMM kernels request their operands with inline-asm loads whose completion hipcc does not
track (hand-counted s_waitcnt).  If the compiler spills the destination register of such a load while it is in flight, the
register is reused and later overwritten by the arriving data -- on the device that showed as a memory access fault.  Every
kernel of those families must therefore compile without VGPR spills and without scratch."""
import os, re, shutil, subprocess, tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "chatterbox-vllm2_amd", "csrc", "t3_kernels.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ASM_LOAD_FAMILIES = ("gemm2_kernel", "gemm2_loop_kernel", "pgemm_kernel")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")
def test_asm_load_kernels_do_not_spill():
    hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
                            "-c", SRC, "-o", os.path.join(td, "k.o"), "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    name, seen, bad = None, 0, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: .*Name: (\S+)", line)
        if m:
            name = m.group(1); continue
        if name is None or not any(f in name for f in ASM_LOAD_FAMILIES):
            continue
        m = re.search(r"(VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m:
            seen += 1
            if int(m.group(2)) != 0:
                bad.append((name, m.group(1), int(m.group(2))))
    assert seen >= 20, f"resource remarks not found ({seen}): has the remark format changed?"
    assert not bad, f"asm-load kernels with spills / scratch: {bad}"
