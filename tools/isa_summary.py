#!/usr/bin/env python3
"""Compile one .hip file to gfx950 assembly and print, per kernel, the register/scratch budget and
the sequence of memory events, so that serialised `global_load ; s_waitcnt vmcnt(0)` chains and
per-store waits are visible at a glance.

    python tools/isa_summary.py cuda-bundle-adjustment_amd/csrc/kernels/chol_kernels.hip [kernel-substring]

Legend: L global load, S global store, (wN) s_waitcnt vmcnt(N), | s_barrier, b branch, M MFMA,
        d LDS read, D LDS write, [ / ] loop-ish backward branch target is not tracked (read the .s).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.abspath(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    tmp = tempfile.mkdtemp(prefix="isa_")
    inc = os.path.join(ROOT, "cuda-bundle-adjustment_amd")
    contract = "fast" if os.path.basename(src).startswith("chol_") else "off"  # as in the Makefile
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=" + contract, "-Wno-pass-failed",
           "-I", os.path.join(inc, "include"), "-I", os.path.join(ROOT, "include"),
           "--offload-device-only", "-S", src, "-o", os.path.join(tmp, "out.s")]
    subprocess.check_call(cmd, cwd=os.path.dirname(src))
    text = open(os.path.join(tmp, "out.s")).read()
    print("asm:", os.path.join(tmp, "out.s"))
    # split into functions
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pat not in name:
            continue
        ev = []
        for line in body.split("\n"):
            t = line.strip()
            if t.startswith("global_load") or t.startswith("buffer_load") or t.startswith("flat_load"):
                ev.append("L")
            elif t.startswith("global_store") or t.startswith("flat_store"):
                ev.append("S")
            elif t.startswith("s_waitcnt"):
                mm = re.search(r"vmcnt\((\d+)\)", t)
                if mm:
                    ev.append("(w%s)" % mm.group(1))
            elif t.startswith("s_barrier"):
                ev.append("|")
            elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                ev.append("b")
            elif t.startswith("v_mfma"):
                ev.append("M")
            elif t.startswith("ds_read") or t.startswith("ds_bpermute"):
                ev.append("d")
            elif t.startswith("ds_write"):
                ev.append("D")
            elif t.startswith("scratch_"):
                ev.append("x")
        meta = {}
        for key in ("sgpr_count", "vgpr_count", "scratch_en", "private_segment_fixed_size", "group_segment_fixed_size"):
            mm = re.search(r"\.amdhsa_%s\s+(\d+)" % key, text[m.end() - 4000:m.end() + 4000])
            if mm:
                meta[key] = mm.group(1)
        mm = re.search(r"; ScratchSize: (\d+)", body)
        vg = re.search(r"; NumVgprs: (\d+)", text[m.end():m.end() + 3000])
        sc = re.search(r"; ScratchSize: (\d+)", text[m.end():m.end() + 3000])
        oc = re.search(r"; Occupancy: (\d+)", text[m.end():m.end() + 3000])
        # compress runs
        s = "".join(ev)
        s = re.sub(r"(d{4,})", lambda k: "d%d" % len(k.group(1)), s)
        s = re.sub(r"(D{4,})", lambda k: "D%d" % len(k.group(1)), s)
        s = re.sub(r"(M{4,})", lambda k: "M%d" % len(k.group(1)), s)
        print("== %s  vgprs=%s scratch=%s occupancy=%s" % (name, vg and vg.group(1), sc and sc.group(1), oc and oc.group(1)))
        print(s)


if __name__ == "__main__":
    main()
