#!/usr/bin/env python3
"""
EXPERIMENT (round 4), NOT part of the build: keeps the 8-byte instructions of the device code on 8-byte boundaries.

Result (profiles/r04/ab_aligned_fp64.log, tools/ab_bench.py, library built from the rewritten assembly against the plain one):
configs[1] +0.65 %, configs[2] 0.0 %, config 5's shard in float64 +1.2 %.  The penalty this pass removes is real for the
packed-float32 stream of the float32 sampler's trip (which aligns itself); the float64 kernels' VOP3 instructions do not
pay it to any extent worth a second build pipeline.  Kept as the record of how that was established:

  hipcc <flags> --cuda-device-only -S -o dev.s rsf_hip.hip
  python tools/align_encodings.py dev.s dev_al.s --report
  clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c dev_al.s -o dev.o
  lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o dev.out dev.o
  clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=dev.out -output=dev.hipfb
  hipcc <flags> --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang dev.hipfb -shared -o librsf_hip.so rsf_hip.hip

  python tools/align_encodings.py in.s out.s [--report]        (in.s: hipcc --cuda-device-only -S; out.s: to be assembled)

The idea.  On gfx950 a wave issues an 8-byte packed-float32 instruction that starts on a 4-byte boundary one cycle later
than one that starts on an 8-byte boundary (tools/microbench_issue.hip, microbench_trip.hip: 5.07 against 4.06 cycles per
instruction for the same stream), and with one wave per SIMD nothing hides that cycle.  The float64 kernels are almost
entirely VOP3 instructions (v_fma_f64, v_mul_f64, v_add_f64 have no shorter encoding) — but hipcc shrinks every
`d = fma(a, b, d)` to the 4-byte `v_fmac_f64_e32`, a fifth of the hot loop, and each 4-byte instruction flips the phase of
everything behind it: in the sampler's 16-step trip a third to two thirds of the 8-byte instructions start misaligned,
depending on where the trip happens to begin.  Nothing in the source controls encodings, and the backend has no switch for
its shrinking pass; so the assembly is rewritten between hipcc's device compile and the assembler:

  1. `v_fmac_f64_e32 d, a, b`  ->  `v_fma_f64 d, a, b, d`: the same IEEE operation in its 8-byte encoding (not where `a` is
     a literal: VOP3 takes none).  Results are bit-identical by construction.
  2. every basic-block label gets a `.p2align 3` (the assembler pads code with s_nop; the pad is executed on fall-through only);
  3. inside a block the byte phase is followed with the exact sizes llvm-mc reports (-show-encoding), and an `s_nop 0` goes in
     front of an 8-byte instruction that would start on a 4-byte boundary — except inside an s_getpc_b64 sequence, whose
     relocations count bytes from the s_getpc.
Only basic blocks of at least MIN_BLOCK instructions are treated — the unrolled hot loops; aligning all of the code would grow it
by a quarter of a megabyte and push branches of the largest kernel out of their 16-bit range.  The statements of the float32
sampler's generated trip carry their own alignment and are uniform: nothing changes in them.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
MIN_BLOCK = 100     # instructions; shorter blocks are control code
INLINE = re.compile(r"^(-?(0\.5|1\.0|2\.0|4\.0)|-?\d{1,2}|0|[vs]\[\d+:\d+\]|[vs]\d+|vcc|exec)$")


LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
INSTR = re.compile(r"^\s+[a-z]\w*(\s|$)")   # an instruction line of hipcc's output: indented mnemonic (directives start with a dot)


def big_blocks(lines, is_instr):
    """Indices of the lines that belong to basic blocks of at least MIN_BLOCK instructions (blocks are delimited by labels)."""
    big, cur, n = set(), [], 0
    for i, l in enumerate(lines + ["end_of_file:"]):
        if LABEL.match(l):
            if n >= MIN_BLOCK:
                big.update(cur)
            cur, n = [i], 0
        else:
            cur.append(i)
            n += bool(is_instr(l))
    return big


def promote(lines):
    """v_fmac_f64_e32 d, a, b -> v_fma_f64 d, a, b, d (8-byte encoding of the same operation), in the long blocks."""
    big = big_blocks(lines, lambda l: INSTR.match(l) and not l.lstrip().startswith("."))
    n = 0
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(\s*)v_fmac_f64_e32\s+([^,]+),\s*([^,]+),\s*([^;\n]+?)\s*(;.*)?$", l) if i in big else None
        if m and INLINE.match(m.group(3).strip()) and INLINE.match(m.group(4).strip()):
            out.append(f"{m.group(1)}v_fma_f64 {m.group(2).strip()}, {m.group(3).strip()}, {m.group(4).strip()}, {m.group(2).strip()}")
            n += 1
        else:
            out.append(l)
    return out, n


def with_encodings(lines):
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "a.s"), os.path.join(d, "b.s")
        open(src, "w").write("\n".join(lines) + "\n")
        p = subprocess.run([LLVM_MC, "-triple", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-show-encoding", src, "-o", dst], capture_output=True, text=True)
        if p.returncode:
            sys.exit("llvm-mc: " + p.stderr[-2000:])
        return open(dst).read().split("\n")


def align(lines):
    big = big_blocks(lines, lambda l: "; encoding: [" in l)
    out = []
    phase = None       # byte offset mod 8 inside a treated block; None outside
    protect = 0        # instructions left of an s_getpc_b64 sequence
    stats = collections.Counter()
    fn, per_fn = None, collections.Counter()
    for i, l in enumerate(lines):
        m = LABEL.match(l)
        if m:
            name = m.group(1)
            if i in big:
                out.append("\t.p2align\t3")
                stats["labels"] += 1
                phase = 0
            else:
                phase = None
            if not name.startswith("."):
                fn = name
            out.append(l)
            continue
        d = re.match(r"^\s*\.p2align\s+(\d+)", l)
        if d and phase is not None and int(d.group(1)) >= 3:   # (a statement of inline assembly that aligns itself)
            phase = 0
        e = re.search(r"; encoding: \[([^\]]*)\]", l)
        if not e or phase is None:
            out.append(l.split("; encoding:")[0].rstrip() if e else l)
            continue
        size = len(e.group(1).split(","))
        op = l.split()[0]
        if size % 8 == 0 and phase == 4 and not protect:
            out.append("\ts_nop 0")
            stats["nops"] += 1
            per_fn[fn] += 1
            phase = 0
        if size % 8 == 0:
            stats["eight"] += 1
            stats["eight_misaligned"] += phase == 4
        if protect:
            protect -= 1
        if op == "s_getpc_b64":
            protect = 2
        out.append(l.split("; encoding:")[0].rstrip())
        phase = (phase + size) % 8
    return out, stats, per_fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--report", action="store_true")
    args = ap.parse_args()
    lines = open(args.src).read().split("\n")
    lines, n_promoted = promote(lines)
    enc = with_encodings(lines)
    out, stats, per_fn = align([l for l in enc if not l.lstrip().startswith(";   fixup")])
    open(args.dst, "w").write("\n".join(out) + "\n")
    print(f"align_encodings: blocks of >= {MIN_BLOCK} instructions: {stats['labels']}; {n_promoted} v_fmac_f64_e32 -> v_fma_f64; "
          f"{stats['nops']} s_nop inserted; 8-byte instructions in them: {stats['eight']}, still on a 4-byte boundary: {stats['eight_misaligned']}")
    if args.report:
        for f, n in per_fn.most_common(12):
            print(f"   {n:6d} s_nop in {f}")


if __name__ == "__main__":
    main()
