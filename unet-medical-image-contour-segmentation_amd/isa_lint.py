"""Build-time guard for the hand-scheduled kernels of csrc/conv3x3.hip.

conv3x3_fwd_mfma_v2 and conv3x3_wgrad_mfma_v2 issue their vector-memory operations through inline asm
(uh_dma16 / uh_ld16_async / uh_ld8_async) and wait for them with hand-counted `s_waitcnt vmcnt(N)`.  hipcc does
not know that the "+v" destinations of those loads are in flight, so three things would corrupt results
silently: a spill / reload of such a register between issue and wait, a register copy of it (`v_mov`) before
the wait, or a change in the NUMBER of vector-memory instructions between an issue and its wait.  This lint
reads the ISA hipcc produced (`--save-temps`-style `.s`, device side) and fails the build when

  * a guarded kernel has scratch traffic inside its MFMA region (between its first and last `v_mfma`; spills in the rest
    of the enclosing loops are counted and reported, not refused), or
  * any instruction touches the destination registers of an inline-asm load that is still outstanding
    according to an in-order model of `vmcnt` (every vector-memory instruction enters a FIFO at issue,
    `s_waitcnt vmcnt(N)` retires all but the N youngest).

The scan is linear over the function text (loop back edges carry a `vmcnt(0)` + barrier in these kernels, so
the FIFO is empty on both ways into a loop head); it is a lint, not a proof.  The toolchain this source was
validated on is pinned in VALIDATED_ROCM: another version builds with a loud warning and the same checks.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

VALIDATED_ROCM = "7.2.0"
GUARDED = ("conv3x3_fwd_mfma_v2", "conv3x3_wgrad_mfma_v2")

_VMEM = re.compile(r"^(buffer_|global_|scratch_|flat_)(load|store|atomic)")
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def rocm_version() -> str:
    try:
        return open("/opt/rocm/.info/version").read().strip().split("-")[0]
    except OSError:
        return "unknown"


def _regs(text: str) -> set:
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_functions(asm: str) -> Dict[str, List[str]]:
    funcs, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                funcs[name] = cur
                cur = None
            else:
                cur.append(line)
    return funcs


def lint_function(name: str, lines: List[str]) -> Tuple[List[str], dict]:
    """-> (violations, summary) for one kernel."""
    errs = []
    ins = []                      # (text, in_asm_block)
    labels = {}                   # label -> index of the instruction that follows it
    in_asm = False
    for raw in lines:
        s = raw.strip()
        if re.match(r"^\.?[A-Za-z_][\w.$]*:", s) and not s.startswith(";"):
            labels[s.split(":")[0]] = len(ins)
        if s.startswith(";;#ASMSTART") or s.startswith(";#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND") or s.startswith(";#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        ins.append((s.split(";")[0].strip(), in_asm))
    mf = [i for i, (t, _) in enumerate(ins) if t.startswith("v_mfma")]
    # Spills between the first and the last v_mfma are an error (the hand-scheduled stream is not what was written any more).
    # Spills elsewhere in a loop that holds MFMAs (its DMA-issue head, its fence / epilogue tail) cost time once per tile but
    # cannot corrupt anything the second check does not see: they are COUNTED (`scratch_in_enclosing_loops`, reported in the
    # lint json and by build.py) so that a change in register pressure shows up in review.
    scratch_in_region = scratch_in_loops = 0
    if mf:
        lo, hi = mf[0], mf[-1]
        scratch_in_region = sum(1 for t, _ in ins[lo:hi + 1] if t.startswith("scratch_"))
        for i, (t, _) in enumerate(ins):
            m = re.match(r"s_c?branch\w*\s+(\S+)", t)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                a, b = labels[m.group(1)], i
                if any(a <= k <= b for k in mf):
                    lo, hi = min(lo, a), max(hi, b)
        scratch_in_loops = sum(1 for t, _ in ins[lo:hi + 1] if t.startswith("scratch_"))
        if scratch_in_region:
            errs.append(f"{name}: {scratch_in_region} scratch (spill) instructions between the first and the last MFMA")
    # in-order vmcnt model
    fifo: List[Tuple[int, frozenset]] = []        # (instruction index, async destination registers or empty)
    touched = 0
    n_async = 0
    for i, (t, asm_blk) in enumerate(ins):
        m = re.match(r"s_waitcnt\b(.*)", t)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", t)
            if v:
                keep = int(v.group(1))
                fifo = fifo[len(fifo) - keep:] if keep else []
            continue
        pending = set()
        for _, d in fifo:
            pending |= d
        if pending:
            ops = t.split(None, 1)
            used = _regs(ops[1]) if len(ops) > 1 else set()
            hit = used & pending
            if hit:
                touched += 1
                if touched <= 5:
                    errs.append(f"{name}: `{t}` touches v{sorted(hit)} while an inline-asm load into it is outstanding")
        if _VMEM.match(t):
            dst = frozenset()
            if asm_blk and "load" in t.split()[0] and " lds" not in (" " + t):
                ops = t.split(None, 1)[1]
                dst = frozenset(_regs(ops.split(",")[0]))
                n_async += 1
            fifo.append((i, dst))
    return errs, {"mfma": len(mf), "async_loads": n_async, "scratch_in_mfma_region": scratch_in_region,
                  "scratch_in_enclosing_loops": scratch_in_loops, "touches_before_wait": touched}


def lint_asm(asm: str, guarded=GUARDED) -> Tuple[List[str], Dict[str, dict]]:
    errs, report = [], {}
    for name, lines in split_functions(asm).items():
        if not any(g in name for g in guarded):
            continue
        e, s = lint_function(name, lines)
        errs += e
        report[name] = s
    if not report:
        errs.append("isa_lint: no guarded kernel found in the ISA (names changed?)")
    return errs, report


if __name__ == "__main__":
    import json
    import sys
    e, r = lint_asm(open(sys.argv[1]).read())
    print(json.dumps(r, indent=1))
    for x in e:
        print("VIOLATION:", x)
    sys.exit(1 if e else 0)
