"""Build-time hazard check of the inline-asm register loads of the shipped kernels (no GPU).

hipcc neither counts nor tracks an `asm volatile("global_load_dwordx4 %0, ...")`: the destination VGPRs count as written at
;;#ASMEND, so nothing but the author's own hand-counted `s_waitcnt vmcnt(N)` keeps the compiler from reading, copying, spilling or
re-using them while the load is still in flight (round 3: a prefetch load whose destination the compiler re-used landed later and
overwrote a live address -> memory access fault).  This script walks the device ISA of every kernel in program order and keeps the
queue of outstanding vector-memory operations (vmcnt retires them in issue order; compiler loads / stores / LDS-DMA count too):

  * an asm-block `global_load_* vDST, ...` (not the `lds` forms: they have no register destination) enters the queue with its
    destination registers;
  * every `s_waitcnt ... vmcnt(N)` (asm or compiler) retires all but the N youngest entries;
  * VIOLATION 1: any instruction that names a register of a still-outstanding asm destination (a fire-and-forget load's register was
    handed to something else, or a consumer was scheduled above its wait);
  * NOTE (not a violation): `s_endpgm` reached, in textual order, with an asm register load outstanding.  The hardware executes an
    implicit s_waitcnt 0 before s_endpgm, so the data cannot land in another wave's registers; what the note usually shows is a
    conditional issue in a loop's last trip (the looped GEMM asks for "the next group's rows" one group past the end in its 16-wave
    forms: clamped addresses, never consumed).

Control flow is walked linearly (textual order); a backward branch does not reset the queue.  That is conservative for the kernels
here: their asm loads sit in straight-line, fully unrolled code, and the looped GEMM's waits retire everything before its back edge.

Usage: python tools/asm_load_guard.py file.s [...]   (files from `hipcc -save-temps`); exit code 1 on a violation.
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
WAIT = re.compile(r"s_waitcnt\b(.*)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, lines):
    """lines: (line_no, text) of one kernel body.  Returns (n_asm_reg_loads, violations, notes)."""
    in_asm = False
    queue = []          # entries: None (no tracked destination) or (set of dest regs, line_no)
    n_asm = 0
    bad, notes = [], []
    for no, raw in lines:
        t = raw.split(";;#")[0] if ";;#ASM" not in raw else raw
        if ";;#ASMSTART" in raw:
            in_asm = True; continue
        if ";;#ASMEND" in raw:
            in_asm = False; continue
        code = t.split("//")[0].split(" ; ")[0].strip()
        if not code or code.startswith((".", ";")) or code.endswith(":"):
            continue
        w = WAIT.search(code)
        if w:
            m = re.search(r"vmcnt\((\d+)\)", w.group(1))
            if m:
                keep = int(m.group(1))
                if keep < len(queue):
                    queue = queue[len(queue) - keep:] if keep else []
            elif re.fullmatch(r"\s*(0x[0-9a-fA-F]+|\d+)\s*", w.group(1) or ""):
                val = int(w.group(1).strip(), 0)          # raw immediate: vmcnt = bits [3:0] | [15:14] << 4
                keep = (val & 0xF) | (((val >> 14) & 0x3) << 4)
                if keep < len(queue):
                    queue = queue[len(queue) - keep:] if keep else []
            continue
        pending = set().union(*[q[0] for q in queue if q]) if any(queue) else set()
        if VMEM.match(code):
            is_load = code.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))
            to_lds = "_lds_" in code.split()[0] or re.search(r"\blds\b", code) is not None
            ops = code.split(None, 1)[1] if " " in code else ""
            first = ops.split(",")[0]
            srcs = ops[len(first):]
            if pending & regs_of(srcs if (is_load and not to_lds) else ops):
                bad.append((no, "uses a register of an outstanding asm load as an operand", code))
            if in_asm and is_load and not to_lds:
                dst = regs_of(first)
                if pending & dst:
                    bad.append((no, "asm load into a register that is still the destination of an outstanding asm load", code))
                queue.append((dst, no)); n_asm += 1
            else:
                queue.append(None)
            continue
        if code.startswith("s_endpgm"):
            left = [q for q in queue if q]
            if left:
                notes.append((no, f"s_endpgm with {len(left)} asm register load(s) outstanding in textual order (first issued at line {left[0][1]})", code))
            queue = []
            continue
        if pending and (pending & regs_of(code)):
            bad.append((no, "touches a register of an outstanding asm load", code))
    return n_asm, bad, notes


def check_file(path):
    kernels, cur, name = {}, None, None
    for no, raw in enumerate(open(path, errors="replace"), 1):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", raw)
        if m and not raw.startswith((".L", "\t")):
            name = m.group(1); cur = []; kernels[name] = cur; continue
        if cur is not None:
            cur.append((no, raw.rstrip("\n")))
            if ".end_amdhsa_kernel" in raw or raw.strip().startswith(".Lfunc_end"):
                cur = None
    res = {}
    for k, lines in kernels.items():
        n, bad, notes = check_kernel(k, lines)
        if n or bad:
            res[k] = (n, bad, notes)
    return res


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for k, (n, bad, notes) in check_file(p).items():
            print(f"{p}: {k}: {n} asm register loads, {len(bad)} violations, {len(notes)} notes")
            for no, why, code in bad[:8] + notes[:2]:
                print(f"    line {no}: {why}: {code}")
            rc |= bool(bad)
    sys.exit(rc)
