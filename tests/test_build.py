"""Build-time guards (no GPU) over the shipped kernel translation units.

1. The GEMM kernels request their operands with inline-asm loads whose completion hipcc does not track (hand-counted s_waitcnt).  If
   the compiler spills the destination register of such a load while it is in flight, the register is reused and later overwritten by
   the arriving data -- on the device that showed as a memory access fault.  Every kernel of those families must therefore compile
   without VGPR spills and without scratch.
2. tools/asm_load_guard.py walks the device ISA: no instruction may name the destination register of an asm load that no
   `s_waitcnt vmcnt(N)` has retired yet (a fire-and-forget register load, or a consumer scheduled above its wait)."""
import os, re, shutil, subprocess, sys, tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chatterbox-vllm2_amd", "csrc")
UNITS = ("t3_gemm.hip", "t3_attention.hip", "t3_kernels.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ASM_LOAD_FAMILIES = ("gemm2_kernel", "gemm2_loop_kernel", "pgemm_kernel")
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def compiled():
    hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not installed")
    with tempfile.TemporaryDirectory() as td:
        procs = [(u, subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                                       "-fno-fast-math", "-save-temps", "-c", os.path.join(CSRC, u), "-o", u + ".o", "-Rpass-analysis=kernel-resource-usage"],
                                      cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)) for u in UNITS]
        remarks = {}
        for u, p in procs:
            _, err = p.communicate(timeout=900)
            assert p.returncode == 0, err[-2000:]
            remarks[u] = err
        isa = {u: os.path.join(td, u.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s") for u in UNITS}
        for f in isa.values():
            assert os.path.exists(f), f
        yield remarks, isa


def test_asm_load_kernels_do_not_spill(compiled):
    remarks, _ = compiled
    name, seen, bad = None, 0, []
    for line in "\n".join(remarks.values()).splitlines():
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


def test_no_asm_register_load_is_touched_before_its_wait(compiled):
    import asm_load_guard as G
    _, isa = compiled
    total, bad = 0, []
    for u, f in isa.items():
        for k, (n, viol, _notes) in G.check_file(f).items():
            total += n
            bad += [(u, k, *v) for v in viol]
    assert total >= 500, f"only {total} asm register loads found in the shipped ISA: has the asm / ISA text format changed?"
    assert not bad, f"asm register loads touched before their s_waitcnt: {bad[:5]}"


def test_the_guard_sees_a_fire_and_forget_register_load():
    """The checker itself: the round-3 fault pattern (an asm load whose destination the compiler hands to something else while the
    load is in flight), a consumer above its wait, and the clean counted form."""
    import asm_load_guard as G
    def run(body):
        return G.check_kernel("k", list(enumerate(body.strip("\n").split("\n"), 1)))
    reuse = """
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[2:3], off
	;;#ASMEND
	v_mov_b32_e32 v5, 0
	s_waitcnt vmcnt(0)
	s_endpgm
"""
    early = """
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[2:3], off
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[8:11], v[2:3], off nt
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_f32_e32 v0, v4, v8
	s_endpgm
"""
    clean = early.replace("v_add_f32_e32 v0, v4, v8", "v_add_f32_e32 v0, v4, v5\n\ts_waitcnt vmcnt(0)\n\tv_add_f32_e32 v0, v0, v8")
    assert len(run(reuse)[1]) == 1 and len(run(early)[1]) == 1
    n, bad, notes = run(clean)
    assert n == 2 and not bad and not notes
