#!/usr/bin/env python3
"""
The float32 sampler's assembly trip (csrc/rsf_f32_trip.inc) owns v[RSF_F32_TRIP_COMPILER_VGPRS .. 255]: this script compiles
one instantiation of mcmc_f32x2_kernel (no GPU needed) and checks, in its ISA, what that rests on —
  * compiled code (everything outside the asm statements) touches no vector register at or above the private file's base;
  * the kernel uses no AGPR (with AGPRs in use amdgpu_num_vgpr would halve the architectural limit and move them);
  * no scratch access inside the trip loop (loop depth >= 3);
  * every packed instruction of every trip statement begins on an 8-byte boundary (the byte phase is followed through the
    statement: .p2align 3 directives, 4-byte scalar instructions, 8-byte vector ones).

  python tools/check_private_file.py ['mcmc_f32x2_kernel<3, true, false>']        exit status 0 = all hold
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesian-markov-chain-monte-carlo_amd", "csrc")


def main():
    inst = sys.argv[1] if len(sys.argv) > 1 else "mcmc_f32x2_kernel<3, true, false>"
    inc = open(os.path.join(CSRC, "rsf_f32_trip.inc")).read()
    base = int(re.search(r"#define RSF_F32_TRIP_COMPILER_VGPRS (\d+)", inc).group(1))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "k.hip")
        open(src, "w").write(f'#include "{CSRC}/rsf_kernels.h"\nnamespace rsfk {{ template __global__ void {inst}(rsf::Consts, McmcArgs); }}\n')
        p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", os.path.join(d, "k.s"), src],
                           capture_output=True, text=True)
        if p.returncode:
            sys.exit(p.stderr[-2000:])
        text = open(os.path.join(d, "k.s")).read()
    start = text.index("_ZN4rsfk17mcmc_f32x2_kernel")
    lines = text[start:text.index("s_endpgm", start)].split("\n")
    inasm, depth, high, agpr, scratch_in_trip, misaligned, trips = False, 0, [], 0, 0, 0, 0
    for i, l in enumerate(lines):
        if re.match(r"^\.LBB", l):
            depth = 0
        m = re.search(r"Depth=(\d+)", l)
        if m:
            depth = int(m.group(1))
        if "#ASMSTART" in l:
            inasm = True
            body = []
            continue
        if "#ASMEND" in l:
            inasm = False
            if sum(b.startswith("v_pk_") for b in body) > 100:   # a trip statement: follow the byte phase of its instructions
                trips += 1
                phase, bad = None, 0                              # None: unknown until the first .p2align 3
                for b in body:
                    op = b.split()[0]
                    if op == ".p2align":
                        phase = 0 if b.split()[1] == "3" else None
                    elif op.endswith(":"):
                        pass
                    elif op.startswith("s_") or op.endswith("_e32"):   # SOP* and VOP1/VOP2 encodings: 4 bytes
                        phase = None if phase is None else phase ^ 4
                    else:                                          # VOP3 / VOP3P / DS: 8 bytes
                        bad += op.startswith("v_pk_") and phase != 0
                misaligned += bad > 0
            continue
        code = l.split(";")[0].strip()
        if inasm:
            if code:
                body.append(code)
            continue
        if "scratch_" in code and depth >= 3:
            scratch_in_trip += 1
        if re.search(r"\ba\d+\b|a\[\d+:\d+\]|accvgpr", code):
            agpr += 1
        for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", code):
            if int(m.group(2) or m.group(3)) >= base:
                high.append(code)
    ok = not high and not agpr and not scratch_in_trip and trips >= 1 and not misaligned
    print(f"{inst}: private file v[{base}:255]; compiled instructions touching it: {len(high)}; AGPR instructions: {agpr}; "
          f"scratch accesses inside the trip loop: {scratch_in_trip}; trip statements: {trips}, not 8-byte aligned: {misaligned}  ->  {'ok' if ok else 'FAILED'}")
    for h in high[:5]:
        print("   ", h)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
