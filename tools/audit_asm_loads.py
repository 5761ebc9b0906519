"""Audit of the hand-counted asm loads in csrc/conv_ws.hip: between an inline-asm global_load and the inline-asm s_waitcnt that retires
it, NO instruction may read or write the load's destination registers (hipcc does not know the data is still in flight: a register
copy, a spill or a reuse there reads garbage -- cdna_hip_programming.md section 5.7 item 1).  Compiles the file with -save-temps and
scans every convws_kernel instantiation in layout order; prints offending instructions and exits non-zero if there are any.
    python tools/audit_asm_loads.py"""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd", "csrc")
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-slp-vectorize",
                       "-save-temps", "-I", CSRC, "-c", os.path.join(CSRC, "conv_ws.hip"), "-o", os.path.join(tmp, "o.o")], cwd=tmp,
                      stderr=subprocess.DEVNULL)
asm = open(os.path.join(tmp, "conv_ws-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
reg = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in reg.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


bad_total = 0
for m in re.finditer(r"^(_Z13convws_kernel\w+):", asm, re.M):
    name = m.group(1)
    body = asm[m.end():asm.index(".Lfunc_end", m.end())].split("\n")
    inflight, in_asm, bad, nloads, nwaits = set(), False, [], 0, 0
    for i, line in enumerate(body):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if in_asm:
            if t.startswith("global_load"):
                dst = t.split(None, 1)[1].split(",")[0]
                inflight |= regs(dst)
                nloads += 1
            elif t.startswith("s_waitcnt vmcnt"):
                inflight.clear()          # (counted waits retire the OLDEST loads; the kernel waits for every group before using any)
                nwaits += 1
            continue
        if inflight and (regs(t) & inflight) and not t.startswith("s_"):
            bad.append((i, t))
    print("%s: %d asm loads, %d asm waits, %d accesses of in-flight destinations" % (name, nloads, nwaits, len(bad)))
    for b in bad[:10]:
        print("   line %d: %s" % b)
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
