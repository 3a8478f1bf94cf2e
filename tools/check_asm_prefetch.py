#!/usr/bin/env python3
"""Build-time ISA check of the hand-issued .bed prefetch in mxm_fp4_kernel<true> (csrc/corr_build.hip).

The kernel requests packed genotypes two K blocks ahead with inline-assembly `global_load_dwordx4` instructions and
waits for them with counted `s_waitcnt vmcnt(2)` (the compiler's own waits would drain the queue).  The compiler does
not know that the destination VGPRs are still in flight between request and wait: if register allocation ever places
a copy or any other use of those registers in between (it did exactly that to a level-1 experiment whose values were
loop-carried), the kernel computes on stale data without any diagnostic.  This script compiles the translation unit
to device assembly and walks every kernel that contains such loads: from each `global_load_dwordx4 v[a:b]` until the
`s_waitcnt vmcnt(N)` that retires it (loads retire in order: after vmcnt(N) only the N youngest are pending), no
instruction may read or write v[a:b].  Loop back-edges are followed once with the state at the branch.  Exit status 1
on a violation.  usage: check_asm_prefetch.py <file.hip> [hipcc flags...]   (or  --asm file.s)
"""
import re
import subprocess
import sys
import tempfile

KERNEL_PAT = "mxm_fp4_kernelILb1E"


def vregs(tok):
    """set of VGPR numbers named by an operand token: v12, v[4:7]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_function(name, lines):
    """Control-flow walk: every path through the kernel is followed with the list of loads in flight (oldest first);
    a (position, state) pair is visited once, so loops converge."""
    ins_list = []
    labels = {}
    for l in lines:
        l = l.split(";")[0].strip() if not l.startswith(";") else ""
        if not l or l.startswith("."):
            m = re.fullmatch(r"(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = len(ins_list)
            continue
        ins_list.append(l)
    errors, seen_err = [], set()
    nloads = sum(1 for i in ins_list if i.startswith("global_load_dwordx4"))
    visited = set()
    stack = [(0, ())]
    while stack:
        pc, pending = stack.pop()
        pending = list(pending)
        while pc < len(ins_list):
            key = (pc, tuple(pending))
            if key in visited:
                break
            visited.add(key)
            ins = ins_list[pc]
            pc += 1
            op = ins.split()[0]
            if op == "s_endpgm":
                break
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                dst = ins.split(None, 1)[1].split(",")[0]
                busy = frozenset().union(*pending) if pending else frozenset()
                srcs = vregs(ins.split(",", 1)[1]) if "," in ins else set()
                if (srcs | vregs(dst)) & busy and ins not in seen_err:
                    seen_err.add(ins)
                    errors.append(f"{name}: `{ins}` uses a register with a load in flight")
                pending.append(frozenset(vregs(dst)))
                pending = pending[-24:]
                continue
            if op.startswith(("global_store", "global_atomic", "buffer_store", "buffer_atomic", "flat_store", "flat_atomic", "scratch_store")):
                pending.append(frozenset())  # stores and atomics count in vmcnt as well
                pending = pending[-24:]
                continue
            if op == "s_waitcnt":
                vm = re.search(r"vmcnt\((\d+)\)", ins)
                if vm:
                    keep = int(vm.group(1))
                    pending = pending[len(pending) - keep:] if keep else []
                elif not re.search(r"[a-z]+cnt\(", ins):
                    pending = []  # raw immediate form: treat as a full wait
                continue
            busy = frozenset().union(*pending) if pending else frozenset()
            if busy and " " in ins:
                used = vregs(ins.split(None, 1)[1])
                if used & busy and ins not in seen_err:
                    seen_err.add(ins)
                    errors.append(f"{name}: `{ins}` touches v{sorted(used & busy)} while their load is in flight")
            if op == "s_branch":
                pc = labels[ins.split()[-1]]
            elif op.startswith("s_cbranch"):
                stack.append((labels[ins.split()[-1]], tuple(pending)))
    return errors, nloads


def main():
    if sys.argv[1] == "--asm":
        text = open(sys.argv[2]).read()
    else:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            cmd = ["/opt/rocm/bin/hipcc", "--cuda-device-only", "-S", "-Wno-unused-command-line-argument"] + sys.argv[2:] + [sys.argv[1], "-o", tmp.name]
            subprocess.check_call(cmd)
            text = open(tmp.name).read()
    funcs = re.findall(r"^(_Z\w+):\s*;[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, flags=re.S | re.M)
    checked, all_err = 0, []
    for name, body in funcs:
        if KERNEL_PAT not in name:
            continue
        lines = [l.strip() for l in body.splitlines()]
        errs, nl = check_function(name, lines)
        checked += 1
        all_err += errs
        print(f"check_asm_prefetch: {name}: {nl} dwordx4 loads, {len(errs)} violation(s)")
    if not checked:
        print("check_asm_prefetch: kernel not found in the assembly", file=sys.stderr)
        return 1
    for e in all_err[:20]:
        print("  " + e, file=sys.stderr)
    return 1 if all_err else 0


if __name__ == "__main__":
    sys.exit(main())
