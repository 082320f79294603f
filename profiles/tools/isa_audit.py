"""Build-time audit of hand-counted asynchronous register loads in the hand-written kernels (TEST TOOL; runs on the CPU).

Several kernels issue loads through inline asm - `global_load_dwordx4 vN..` in gemm_x3.hip, `ds_read_b128 vN..` in the fused MLP /
attention / convolution kernels - and retire them later with a hand-counted `s_waitcnt`, because hipcc's own wait insertion would drain the
queue (`vmcnt(0)` in front of a load consumed two iterations later).  hipcc treats an asm statement's output registers as written when
the statement ends, so nothing but the wait statement's in/out operands keeps it from reading, copying (e.g. VGPR -> AGPR), spilling or
re-using such a register while its data is still in flight (cdna_hip_programming.md, section 5.7, item 1: "forms (ii)/(iii) pin order, not
register allocation").  Whether it did is a property of the generated code - so it is checked on the generated code:

    for every kernel in the `-S` output of a source file, a forward data-flow pass over its basic blocks tracks the set of registers that
    are the destination of an asm load and have not yet been named by a retiring wait; any instruction that reads or writes such a
    register in between is a violation.

A retiring wait is an asm `s_waitcnt` that lists the registers it retires behind the instruction (`; landed v[4:7] v[8:11]`: the operands
are printed by the asm statement itself, so the list is what the register allocator really assigned), or any `s_waitcnt` with
`vmcnt(0)` / `lgkmcnt(0)` (retires every load of that queue).  Waits without a list and with a non-zero count retire nothing here
(conservative).  A load whose destination is still in flight when the kernel ends, or that is never retired by name, is reported too.

Kernels whose counted waits do not name registers (the bf16-mode kernels) are covered by a second analysis that simulates the two memory
queues instead (`--counting`, audit_function_counting below).

    python profiles/tools/isa_audit.py mastermetastyletransfer_amd/csrc/gemm_x3.hip [more.hip ...]      (uses build.py's flags per file)
    python profiles/tools/isa_audit.py --counting mastermetastyletransfer_amd/csrc/mlp_bf16.hip ...
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from typing import Dict, List, Set, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

REG_RE = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
LOAD_RE = re.compile(r"^(global_load_\w+|buffer_load_\w+|flat_load_\w+|scratch_load_\w+|ds_read\w*|ds_load\w*)\s+([va](?:\[\d+:\d+\]|\d+))")
LABEL_RE = re.compile(r"^(\.?[A-Za-z_$][\w$.]*):")
BRANCH_RE = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\.?[\w$.]+)")


def regs_of(text: str) -> Set[Tuple[str, int]]:
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


class Instr:
    __slots__ = ("text", "comment", "in_asm", "line")

    def __init__(self, text, comment, in_asm, line):
        self.text, self.comment, self.in_asm, self.line = text, comment, in_asm, line


def parse_functions(asm_text: str) -> Dict[str, List[object]]:
    """kernel name -> list of ('label', name) / Instr, in order"""
    funcs: Dict[str, List[object]] = {}
    cur, name, in_asm = None, None, False
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.strip()
        if not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = LABEL_RE.match(line)
        if m and not raw[0].isspace():
            lab = m.group(1)
            if lab.startswith(".Lfunc_end"):
                cur, name = None, None
            elif not lab.startswith("."):                      # a function symbol
                name, cur = lab, []
                funcs[name] = cur
            elif cur is not None:
                cur.append(("label", lab))
            continue
        if cur is None or line.startswith(".") or line.startswith(";"):
            continue
        code, _, comment = line.partition(";")
        code = code.strip()
        if code:
            cur.append(Instr(code, comment.strip(), in_asm, ln))
    return {k: v for k, v in funcs.items() if any(isinstance(x, Instr) for x in v)}


def _blocks_of(items: List[object]):
    """basic blocks and successor lists of a kernel listing (shared by both analyses)"""
    blocks: List[List[Instr]] = [[]]
    label_block: Dict[str, int] = {}
    for it in items:
        if isinstance(it, tuple):
            if blocks[-1]:
                blocks.append([])
            label_block[it[1]] = len(blocks) - 1
        else:
            blocks[-1].append(it)
            if BRANCH_RE.match(it.text) or it.text.startswith(("s_endpgm", "s_setpc")):
                blocks.append([])
    succ: List[List[int]] = []
    for i, b in enumerate(blocks):
        s_ = []
        last = b[-1] if b else None
        m = BRANCH_RE.match(last.text) if last else None
        if m:
            if m.group(2) in label_block:
                s_.append(label_block[m.group(2)])
            if m.group(1) != "s_branch" and i + 1 < len(blocks):
                s_.append(i + 1)
        elif not (last and last.text.startswith(("s_endpgm", "s_setpc"))) and i + 1 < len(blocks):
            s_.append(i + 1)
        succ.append(s_)
    return blocks, succ


def audit_function(items: List[object]) -> Tuple[List[str], int, int]:
    """-> (violations, asm loads seen, named retirements seen)"""
    blocks, succ = _blocks_of(items)

    n_loads = n_named = 0

    def transfer(state: Dict[Tuple[str, int], str], b: List[Instr], report):
        """state: register -> queue ('vm' / 'lgkm') of the in-flight asm load that targets it"""
        nonlocal n_loads, n_named
        st = dict(state)
        for ins in b:
            t = ins.text
            if t.startswith("s_waitcnt"):
                if "landed" in ins.comment:
                    for r in regs_of(ins.comment):
                        st.pop(r, None)
                    if report is not None:
                        n_named += 1
                if re.search(r"vmcnt\(0\)", t):
                    st = {r: q for r, q in st.items() if q != "vm"}
                if re.search(r"lgkmcnt\(0\)", t):
                    st = {r: q for r, q in st.items() if q != "lgkm"}
                continue
            m = LOAD_RE.match(t) if ins.in_asm else None
            if m:
                dst = regs_of(m.group(2))
                others = regs_of(t[m.end():])
                bad = others & st.keys()
                if bad and report is not None:
                    report(ins, bad)
                q = "lgkm" if m.group(1).startswith("ds_") else "vm"
                for r in dst:
                    st[r] = q
                if report is not None:
                    n_loads += 1
                continue
            if st:
                bad = regs_of(t) & st.keys()
                if bad and report is not None:
                    report(ins, bad)
        return st

    # forward data flow, union at joins, to a fixed point
    inn: List[Dict] = [dict() for _ in blocks]
    work = list(range(len(blocks)))
    while work:
        i = work.pop(0)
        out = transfer(inn[i], blocks[i], None)
        for j in succ[i]:
            merged = dict(inn[j])
            merged.update(out)
            if merged != inn[j]:
                inn[j] = merged
                if j not in work:
                    work.append(j)
    violations: List[str] = []

    def report(ins: Instr, bad):
        regs = " ".join(f"{k}{i}" for k, i in sorted(bad))
        violations.append(f"line {ins.line}: `{ins.text}` touches {regs} while an asm load into it is still in flight")

    for i, b in enumerate(blocks):
        out = transfer(inn[i], b, report)
        if b and b[-1].text.startswith("s_endpgm") and out:
            regs = " ".join(f"{k}{i}" for k, i in sorted(out))
            violations.append(f"line {b[-1].line}: kernel ends with asm loads into {regs} never retired by a naming wait")
    return violations, n_loads, n_named


VM_RE = re.compile(r"^(global_|buffer_|flat_|scratch_)")
LGKM_RE = re.compile(r"^(ds_|s_load_|s_buffer_load|s_memtime|s_memrealtime|s_sendmsg|s_dcache)")


def audit_function_counting(items: List[object]) -> Tuple[List[str], int]:
    """The same question for kernels whose counted waits do NOT name registers (the bf16-mode kernels: mlp_bf16, mlp_c128_bf16,
    linear_ts_bf16, conv_ring_bf16, attn_block_bf16, window_attn_proj_bf16): the two memory queues are simulated instead.  Every
    vector-memory operation (asm or compiler-issued; loads, stores, LDS-DMA) enters the vm queue, every LDS / scalar-memory operation the
    lgkm queue, in program order; `s_waitcnt vmcnt(N)` / `lgkmcnt(N)` retires all but the N youngest (operations of one queue complete
    in order on this ISA; scalar loads may not, which only matters if one is issued next to asm reads in flight: reported).  The
    destination of an ASM load is in flight until it is retired; any instruction that touches it before is a violation.
    A queue state is the sequence of operations from the oldest asm load still in flight to the youngest operation (older ones cannot
    change which asm loads a count retires).  States flow over the kernel's basic blocks to a fixed point; where two paths meet with
    DIFFERENT states - a branch around a memory operation while asm loads are in flight - the join is reported as not analysable
    instead of being guessed at.  -> (violations, asm loads seen)"""
    blocks, succ = _blocks_of(items)

    def norm(q):
        for i, e in enumerate(q):
            if e:
                return tuple(q[i:])
        return ()

    n_loads = 0

    def transfer(state, b: List[Instr], viol):
        nonlocal n_loads
        vmq, lgq = list(state[0]), list(state[1])
        for ins in b:
            t = ins.text
            if t.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    del vmq[: max(0, len(vmq) - int(m.group(1)))]
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    del lgq[: max(0, len(lgq) - int(m.group(1)))]
                continue
            fl = set().union(*vmq, *lgq) if (vmq or lgq) else set()
            m = LOAD_RE.match(t) if ins.in_asm else None
            if m:
                bad = regs_of(t[m.end():]) & fl
                if bad and viol is not None:
                    viol.append(f"line {ins.line}: `{t}` reads " + " ".join(f"{k}{i}" for k, i in sorted(bad)) + " while an asm load into it is in flight")
                (lgq if m.group(1).startswith("ds_") else vmq).append(frozenset(regs_of(m.group(2))))
                if viol is not None:
                    n_loads += 1
                continue
            if fl:
                bad = regs_of(t) & fl
                if bad and viol is not None:
                    viol.append(f"line {ins.line}: `{t}` touches " + " ".join(f"{k}{i}" for k, i in sorted(bad)) + " while an asm load into it is in flight")
                if viol is not None and t.startswith(("s_load_", "s_buffer_load", "s_memtime", "s_memrealtime")) and any(lgq):
                    viol.append(f"line {ins.line}: `{t}` (scalar memory, may complete out of order) issued while asm LDS reads are in flight")
            if VM_RE.match(t):
                vmq.append(frozenset())
            elif LGKM_RE.match(t):
                lgq.append(frozenset())
        return (norm(vmq), norm(lgq))

    viol: List[str] = []
    inn: List[object] = [None] * len(blocks)
    inn[0] = ((), ())
    work = [0]
    joins_reported = set()
    guard = 0
    while work and guard < 100000:
        guard += 1
        i = work.pop(0)
        out = transfer(inn[i], blocks[i], None)
        for j in succ[i]:
            if inn[j] is None:
                inn[j] = out
                work.append(j)
            elif inn[j] != out:
                if j not in joins_reported:
                    joins_reported.add(j)
                    first = blocks[j][0].line if blocks[j] else -1
                    viol.append(f"line {first}: paths meet with different asm loads in flight / different queue contents behind them (not analysable)")
    for i, b in enumerate(blocks):
        if inn[i] is not None:
            out = transfer(inn[i], b, viol)
            if b and b[-1].text.startswith("s_endpgm") and (out[0] or out[1]):
                viol.append(f"line {b[-1].line}: kernel ends with asm loads in flight")
    return viol, n_loads


def audit_source_counting(src: str, extra: List[str] = ()) -> Tuple[Dict[str, List[str]], int]:
    funcs = parse_functions(compile_to_asm(src, extra))
    res, loads = {}, 0
    for name, items in funcs.items():
        v, nl = audit_function_counting(items)
        loads += nl
        if v:
            res[name] = v
    return res, loads


def build_flags(src: str) -> List[str]:
    sys.path.insert(0, ROOT)
    from mastermetastyletransfer_amd import build as B
    return [*B.FLAGS, *B.EXTRA_FLAGS.get(os.path.basename(src), [])]


def compile_to_asm(src: str, extra: List[str] = ()) -> str:
    sys.path.insert(0, ROOT)
    from mastermetastyletransfer_amd import build as B
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [B.HIPCC, *[f for f in build_flags(src) if f != "-fPIC"], *extra, "-S", "--offload-device-only", src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed:\n{' '.join(cmd)}\n{r.stderr}")
        return open(out).read()


def audit_source(src: str, extra: List[str] = ()) -> Tuple[Dict[str, List[str]], int, int]:
    funcs = parse_functions(compile_to_asm(src, extra))
    res, loads, named = {}, 0, 0
    for name, items in funcs.items():
        v, nl, nn = audit_function(items)
        loads += nl
        named += nn
        if v:
            res[name] = v
    return res, loads, named


def main():
    rc = 0
    counting = "--counting" in sys.argv
    for src in [a for a in sys.argv[1:] if not a.startswith("--")]:
        if counting:
            res, loads = audit_source_counting(src)
            named = 0
        else:
            res, loads, named = audit_source(src)
        print(f"{os.path.basename(src)}: {loads} asm register loads, {named} naming waits, {sum(len(v) for v in res.values())} violations")
        for name, v in res.items():
            rc = 1
            print(f"  {name}")
            for line in v[:20]:
                print("    " + line)
    sys.exit(rc)


if __name__ == "__main__":
    main()
