#!/usr/bin/env python3
"""Skeleton view of one kernel's gfx950 ISA: every vector-memory instruction, every `s_waitcnt vmcnt`, every barrier and branch, in program
order, with the instruction mix of what lies between them -- a load whose `s_waitcnt vmcnt(0)` follows it directly is an exposed memory round
trip, thirty-two `s_cbranch_execz` in a row are a per-element conditional the compiler turned into exec-masked blocks.
Also prints the kernel's static mix (VALU / MFMA / LDS / scalar-spill `v_readlane` / `v_writelane` counts).
usage: tools/isa_skeleton.py <object with embedded gfx950 code, e.g. alignnet-3d_amd/csrc/alignnet_train.o> <substring of the (mangled or demangled) kernel name> [--full]"""
import collections
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from inflight_scan import disassemble  # noqa: E402


def functions(text):
    cur, out = None, []
    lines = text.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m:
            if cur:
                out.append((cur[0], lines[cur[1] + 1:i]))
            cur = (m.group(1), i)
    if cur:
        out.append((cur[0], lines[cur[1] + 1:]))
    return out


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    funs = functions(disassemble(obj))
    names = subprocess.run(["c++filt"], input="\n".join(f[0] for f in funs), capture_output=True, text=True).stdout.split("\n")
    hits = [(dn, body) for (n, body), dn in zip(funs, names) if pat in n or pat in dn]
    if not hits:
        raise SystemExit("no kernel matches %r; kernels: %s" % (pat, ", ".join(sorted(set(d.split("(")[0] for d in names if d))[:40])))
    for dn, body in hits:
        body = [l.split("//")[0].strip() for l in body]
        mix = collections.Counter()
        for l in body:
            op = (l.split() or [""])[0]
            if op.startswith("v_mfma"): mix["mfma"] += 1
            elif op in ("v_readlane_b32", "v_writelane_b32"): mix["sgpr spill moves"] += 1
            elif op.startswith("v_"): mix["valu"] += 1
            elif op.startswith("ds_"): mix["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")): mix["vmem"] += 1
            elif op.startswith("scratch_"): mix["scratch"] += 1
            elif op.startswith("s_cbranch_exec"): mix["exec branches"] += 1
            elif op.startswith("s_"): mix["salu"] += 1
        print("== %s\n   %d instructions: %s" % (dn[:140], len([l for l in body if l]), ", ".join("%s %d" % kv for kv in mix.most_common())))
        cnt = collections.Counter()

        def flush():
            if sum(cnt.values()) > (0 if "--full" in sys.argv else 12):
                print("        ... valu %d salu %d lds %d mfma %d" % (cnt["v"], cnt["s"], cnt["ds"], cnt["mfma"]))
            cnt.clear()
        for i, l in enumerate(body):
            op = (l.split() or [""])[0]
            if not op:
                continue
            if op.startswith(("global_", "buffer_", "flat_", "scratch_")) or op == "s_barrier" or (op == "s_waitcnt" and "vmcnt" in l) or op.startswith("s_cbranch") or op == "s_branch":
                flush()
                print("%6d  %s" % (i, l[:110]))
            elif op.startswith("v_mfma"): cnt["mfma"] += 1
            elif op.startswith("v_"): cnt["v"] += 1
            elif op.startswith("ds_"): cnt["ds"] += 1
            else: cnt["s"] += 1
        flush()


if __name__ == "__main__":
    main()
