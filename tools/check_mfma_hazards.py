#!/usr/bin/env python3
"""Scan the gfx950 ISA of the HIP library for an MFMA result that is consumed by a non-MFMA instruction
across a basic-block boundary with too few wait states in between.

Why: inside one basic block the compiler pads MFMA -> VALU reads with s_nop; a label / conditional branch
between the MFMA chain and the first v_accvgpr_read escaped that padding in this toolchain (round 1: a run-time
`if (timing)` stage timer after the layer-5 MFMAs of ae_bwd_kernel made accumulator element 3 miss the last
k-step, timing-dependently).  The product kernels therefore keep MFMA chains and the first read of their result
in one block; this script is the regression check (tests/test_abi_and_host.py runs it on the built library's source).

    python tools/check_mfma_hazards.py            # compiles csrc/st_api.hip to ISA and scans every kernel
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "signaltrain_amd", "csrc", "st_api.hip")
# wait states an independent VALU read needs after the MFMA issues (passes + 3, rounded up generously)
NEED = {"16x16x4_": 10, "32x32x2_": 18, "4x4": 4,     # = what the compiler pads to inside one block (s_nop 9 / s_nop 15 + 2)
        "16x16x16": 8}                               # v_mfma_f32_16x16x16_{bf16,f16}: the smallest in-block MFMA -> VALU distance the compiler emits anywhere in the
                                                     # library is 8 (measured over all 219 kernels, round 3); other types: 11, generously


def regs(tok):
    """'a[0:3]' -> {'a0',..,'a3'}; 'v17' -> {'v17'}"""
    m = re.fullmatch(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([av])(\d+)", tok)
    return {f"{m.group(1)}{m.group(2)}"} if m else set()


def parse(lines):
    ins, labels = [], {}
    for ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = len(ins)
            continue
        if ln.endswith(":"):
            continue
        op, _, rest = ln.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        ins.append((op, ops))
    return ins, labels


def scan_kernel(name, lines):
    ins, labels = parse(lines)
    bad = []
    for i, (op, ops) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        need = next((v for k, v in NEED.items() if k in op), 11)
        dst = regs(ops[0])
        # walk forward over all paths for `need` wait states
        work, seen = [(i + 1, 0, False)], set()
        while work:
            j, dist, crossed = work.pop()
            while j < len(ins) and dist < need:
                if (j, crossed) in seen:
                    break
                seen.add((j, crossed))
                o2, p2 = ins[j]
                if any(idx == j for idx in labels.values()):
                    crossed = True
                if o2.startswith("v_mfma"):
                    # a dependent MFMA interlocks in hardware; an overwriting one ends this def's life
                    if regs(p2[0]) & dst:
                        break
                elif o2 == "s_nop":
                    dist += int(p2[0])
                elif o2.startswith("s_cbranch") or o2 == "s_branch":
                    tgt = labels.get(p2[0])
                    if tgt is not None:
                        work.append((tgt, dist + 1, True))
                    if o2 == "s_branch":
                        break
                    crossed = True
                elif o2 in ("s_endpgm",):
                    break
                else:
                    srcs = set().union(*[regs(t) for t in (p2[1:] if not o2.startswith(("global_store", "ds_write", "scratch_store")) else p2)]) if p2 else set()
                    if srcs & dst and crossed:
                        bad.append((name, i, j, op, o2, sorted(srcs & dst)[:2], dist))
                        break
                    if p2 and regs(p2[0]) & dst and not o2.startswith(("global_store", "ds_write", "scratch_store")):
                        break   # overwritten
                dist += 1
                j += 1
    return bad


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "st.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",      # the Makefile's code generation flags
               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, SRC]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur, name = [], None, None
    for ln in text:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
        elif ln.startswith("\t.amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur.append(ln)
    total = []
    for name, lines in kernels:
        total += scan_kernel(name, lines)
    for b in total[:40]:
        print("HAZARD %s: mfma@%d -> %s@%d reads %s after %d wait states across a block boundary" % (b[0][:60], b[1], b[4], b[2], b[5], b[6]))
    print(f"{len(kernels)} kernels scanned, {len(total)} cross-block MFMA result hazards")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
