"""Structural check of the built device code: no instruction writes a VGPR that is the destination of a global load still in flight.

Why: csrc/kernels_infer.h (mfma_rows) issues its weight stream by hand -- `global_load_dwordx4` inside asm statements, retired by counted
`s_waitcnt vmcnt(N)` asm statements.  The compiler does not know those loads are outstanding, so nothing but the C++ source's data flow
keeps it from handing one of their destination registers to a new value before the wait (round 3: `v_mov_b32 v142, 0` scheduled in
front of the final drain while `global_load_dwordx4 v[142:145]` was in flight -- the arg-max counter of train_fwd_phase3_wide<true, true>
became a weight; nothing was spilled, so the spill-free assertion did not see it).  This scan sees it in the ISA.

Model (conservative for memory ops the scan knows, linear over each function): vector-memory operations retire in order; `s_waitcnt vmcnt(N)`
leaves at most the N newest outstanding.  Loads carry their destination registers; stores / atomics without return only count.  Any
non-memory instruction whose destination overlaps an outstanding load's destination is reported.

usage: inflight_scan.py <object with an embedded gfx950 code object | extracted code object> [name filter]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
VM_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load)")
VM_OTHER = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic|flat_atomic)")
NO_VDST = re.compile(r"^(ds_write|ds_add|ds_sub|s_|v_cmp|v_cmpx|global_store|buffer_store|flat_store|scratch_store|v_nop|ds_nop|buffer_wbl2|buffer_inv)")
REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def vrange(op):
    m = REG.match(op.strip())
    if not m:
        return None
    if m.group(1) is not None:
        return int(m.group(1)), int(m.group(1))
    return int(m.group(2)), int(m.group(3))


def disassemble(path):
    with tempfile.TemporaryDirectory() as tmp:
        src = path
        if not open(path, "rb").read(20).startswith(b"\x7fELF") or b"amdgcn" not in open(path, "rb").read(4096) or True:
            # host object / shared library with embedded bundles: extract them next to a copy
            cp = os.path.join(tmp, "x.o")
            with open(path, "rb") as f, open(cp, "wb") as g:
                g.write(f.read())
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            cand = [os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n]
            src = cand[0] if cand else path
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", src], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()


def scan(text, name_filter=None):
    """-> (functions scanned, loads followed, [(function, line, instruction, clobbered load)]).
    Every vector load is followed forward through STRAIGHT-LINE code (the walk stops, inconclusive, at a branch or the end of the function) until a
    `s_waitcnt vmcnt(N)` retires it (N < the number of vector-memory operations issued after it, itself excluded ... in-order return); any instruction
    in between whose destination overlaps the load's destination is reported.  No control-flow false positives: both instructions lie in one block."""
    findings, nfun, nload = [], 0, 0
    funs, cur = [], None
    for ln, line in enumerate(text.splitlines()):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = (m.group(1), [])
            if name_filter is None or name_filter in cur[0]:
                funs.append(cur)
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if ins:
            cur[1].append((ln + 1, ins))
    for fun, body in funs:
        nfun += 1
        for i, (ln, ins) in enumerate(body):
            mn, _, rest = ins.partition(" ")
            if not VM_LOAD.match(mn) or " lds" in ins:
                continue
            dst = vrange(rest.split(",")[0])
            if dst is None:
                continue
            nload += 1
            younger = 0
            for ln2, ins2 in body[i + 1:]:
                mn2, _, rest2 = ins2.partition(" ")
                if mn2.startswith("s_cbranch") or mn2 in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                    break
                if mn2 == "s_waitcnt":
                    w = re.search(r"vmcnt\((\d+)\)", ins2)
                    if w and int(w.group(1)) <= younger:
                        break
                    continue
                if VM_LOAD.match(mn2) or VM_OTHER.match(mn2):
                    younger += 1
                    continue   # (a younger load into the same registers returns after the older one: in-order, benign)
                elif NO_VDST.match(mn2) or not rest2:
                    continue
                d = vrange(rest2.split(",")[0])
                if d is not None and d[0] <= dst[1] and dst[0] <= d[1]:
                    findings.append((fun, ln2, ins2, ins))
    return nfun, nload, findings


if __name__ == "__main__":
    txt = disassemble(sys.argv[1])
    nfun, nload, f = scan(txt, sys.argv[2] if len(sys.argv) > 2 else None)
    print("%d functions, %d vector loads followed, %d writes to a register with a load in flight" % (nfun, nload, len(f)))
    for fun, ln, ins, ltxt in f[:40]:
        print("  %s: line %d: `%s` while `%s` is outstanding" % (fun[:70], ln, ins, ltxt))
    sys.exit(1 if f else 0)
