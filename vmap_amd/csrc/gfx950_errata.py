"""Build-time pass over the compiler's gfx950 assembly (run by __graft_entry__.build() for every translation unit of the library).

WHY.  Measured on MI355X in round 5 (tests/tools/pk_form_probe.hip, profiles/round5i_pk_form_probe.jsonl; HISTORY.md, round 5):
a packed-float32 instruction - v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 - whose LOW result takes src0's low register and src1's
HIGH register (`op_sel:[0,1]`, any op_sel_hi) returns a wrong low result in lanes 48..63 (src1's high register read as zero), about
once per 300-1000 executions, WHILE ANOTHER WAVE ON THE SAME SIMD runs the usual matrix-instruction -> VALU -> LDS mix.  Alone on
the SIMD, or next to matrix instructions / LDS / VALU work taken one at a time, it never fails; the other fifteen op_sel / op_sel_hi
combinations, the third source's selections and v_pk_mov_b32 never fail; wait states do not matter.  hipcc 7.2's SLP vectoriser
emits the form freely (76 instances in this library's seven units), among them in the kernels that run two workgroups per CU.

WHAT.  Every unit's device code goes   hipcc --cuda-device-only -S  ->  rewrite()  ->  assembler  ->  lld  ->  offload bundle,   and
the host side is compiled with that bundle embedded (-fcuda-include-gpubinary): the steps hipcc's own driver runs (`hipcc -###
-save-temps`), with one pass in the middle.  rewrite() swaps the first two sources of an affected instruction - multiplication,
addition and the product of a fused multiply-add are commutative bit for bit - which turns `op_sel:[0,1]` into `op_sel:[1,0]`, a
form that never failed.  risky() is the lint: build() refuses to ship a unit in which it still finds the form (also in instructions
this pass does not know how to rewrite), and tests/test_code_objects.py disassembles the SHIPPED library and applies it again.
"""
from __future__ import annotations

import os
import re
import subprocess

LLVM_BIN = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
OFFLOAD_TARGETS = "host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950"

_PK = re.compile(r"^(\s*)(v_pk_[a-z0-9_]+)\s+(.*)$")
_MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]")
_FIRST_MOD = re.compile(r"\s(op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b")
COMMUTATIVE = ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")      # in their first two sources
UNAFFECTED = ("v_pk_mov_b32",)        # low result = src0[op_sel[0]] only: src1 never feeds the low half (and measured: never wrong)


def _parse(line):
    """(indent, mnemonic, [operands], modifier text, comment) of a packed instruction, or None."""
    m = _PK.match(line)
    if not m:
        return None
    indent, op, rest = m.groups()
    rest, sep, comment = rest.partition(";")
    rest = rest.rstrip()
    mm = _FIRST_MOD.search(rest)
    ops, mods = (rest[:mm.start()], rest[mm.start():]) if mm else (rest, "")
    return indent, op, [o.strip() for o in ops.split(",")], mods, (sep + comment).rstrip("\n")


def _low_half_takes_src0_lo_src1_hi(mods):
    sel = dict(_MOD.findall(mods)).get("op_sel")
    if not sel:
        return False
    s = sel.split(",")
    return len(s) >= 2 and s[0] == "0" and s[1] == "1"


def rewrite(text):
    """The assembly with every affected commutative instruction's first two sources swapped; returns (text, instructions rewritten)."""
    out, n = [], 0
    for line in text.splitlines(keepends=True):
        p = _parse(line)
        if p and p[1] in COMMUTATIVE and _low_half_takes_src0_lo_src1_hi(p[3]):
            indent, op, ops, mods, comment = p
            ops[1], ops[2] = ops[2], ops[1]                      # ops[0] is the destination

            def swap(mo):
                v = mo.group(2).split(",")
                v[0], v[1] = v[1], v[0]
                return f"{mo.group(1)}:[{','.join(v)}]"
            line = f"{indent}{op} {', '.join(ops)}{_MOD.sub(swap, mods)}{(' ' + comment) if comment else ''}\n"
            n += 1
        out.append(line)
    return "".join(out), n


def risky(text):
    """Lines of an assembly / disassembly listing that still hold the affected form (any packed instruction not known to be immune)."""
    bad = []
    for line in text.splitlines():
        line = re.sub(r"\s*//.*$", "", line)                    # llvm-objdump appends the encoding as a // comment
        p = _parse(line)
        if p and p[1] not in UNAFFECTED and _low_half_takes_src0_lo_src1_hi(p[3]):
            bad.append(line.strip())
    return bad


def compile_unit(hipcc, flags, include_dir, src, obj, log=None):
    """src (.hip) -> obj (host object with the rewritten device code embedded).  Returns the number of instructions rewritten."""
    stem = obj[:-2] if obj.endswith(".o") else obj
    asm, fixed, dev_o, dev_out, fatbin = (stem + e for e in (".dev.s", ".dev.fixed.s", ".dev.o", ".dev.out", ".hipfb"))
    run = lambda cmd: subprocess.run(cmd, check=True, stdout=log, stderr=log)
    run([hipcc] + flags + ["--cuda-device-only", "-S", "-I", include_dir, src, "-o", asm])
    with open(asm) as fh:
        text, n = rewrite(fh.read())
    left = risky(text)
    if left:
        raise RuntimeError(f"{src}: {len(left)} packed instructions with op_sel:[0,1] that the erratum pass cannot rewrite: {left[:3]}")
    with open(fixed, "w") as fh:
        fh.write(text)
    run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", fixed, "-o", dev_o])
    run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_out, dev_o])
    run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096", f"-targets={OFFLOAD_TARGETS}",
         "-input=/dev/null", f"-input={dev_out}", f"-output={fatbin}"])
    run([hipcc] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-c", "-I", include_dir, src, "-o", obj])
    for f in (asm, fixed, dev_o):
        os.remove(f)
    return n


def disassemble_library(path, workdir):
    """The gfx950 code objects embedded in a linked shared object (one per translation unit), disassembled: [(name, text)]."""
    os.makedirs(workdir, exist_ok=True)
    fat = os.path.join(workdir, "lib.hip_fatbin")
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for i, s in enumerate(starts):
        one = os.path.join(workdir, f"bundle{i}.hipfb")
        with open(one, "wb") as fh:
            fh.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, f"unit{i}.co")
        subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={one}", f"--output={co}"], check=True)
        dis = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True)
        out.append((f"unit{i}", dis.stdout))
    return out


if __name__ == "__main__":
    import sys
    text, n = rewrite(open(sys.argv[1]).read())
    sys.stdout.write(text)
    print(f"rewritten: {n}; still risky: {len(risky(text))}", file=sys.stderr)
