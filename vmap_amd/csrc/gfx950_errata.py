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


def _llvm_bin():
    """Directory of the LLVM tools: $ROCM_LLVM_BIN, else where hipcc's driver says its clang lives, else /opt/rocm/lib/llvm/bin."""
    if os.environ.get("ROCM_LLVM_BIN"):
        return os.environ["ROCM_LLVM_BIN"]
    try:
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--print-prog-name=clang"], capture_output=True, text=True, timeout=60)
        if r.returncode == 0 and os.path.isabs(r.stdout.strip()) and os.path.exists(r.stdout.strip()):
            return os.path.dirname(r.stdout.strip())
    except Exception:      # noqa: BLE001
        pass
    return "/opt/rocm/lib/llvm/bin"


LLVM_BIN = _llvm_bin()
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


def tools(hipcc):
    """{name: path} of the LLVM tools the pass drives besides hipcc.  Order: $ROCM_LLVM_BIN (override), the directory hipcc's own
    driver reports (`hipcc --print-prog-name=clang`), /opt/rocm/lib/llvm/bin.  Raises with a message naming what is missing."""
    names = ("clang", "lld", "clang-offload-bundler")
    dirs = []
    if os.environ.get("ROCM_LLVM_BIN"):
        dirs.append(os.environ["ROCM_LLVM_BIN"])
    try:
        r = subprocess.run([hipcc, "--print-prog-name=clang"], capture_output=True, text=True, timeout=60)
        if r.returncode == 0 and os.path.isabs(r.stdout.strip()):
            dirs.append(os.path.dirname(r.stdout.strip()))
    except Exception:
        pass
    dirs.append("/opt/rocm/lib/llvm/bin")
    for d in dirs:
        found = {n: os.path.join(d, n) for n in names}
        if all(os.path.exists(f) for f in found.values()):
            return found
    raise FileNotFoundError(f"gfx950_errata: need {names} next to hipcc's clang; looked in {dirs} (set ROCM_LLVM_BIN)")


def hipcc_version(hipcc):
    try:
        r = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60)
        return [l.strip() for l in r.stdout.splitlines() if l.strip()][:2]
    except Exception as e:      # noqa: BLE001
        return [f"unknown ({type(e).__name__})"]


NO_SLP = "-fno-slp-vectorize"      # removes the form at its source (the SLP vectoriser emits it); costs configs[4] 14 %, the headline 0.7 %
                                   # (profiles/round5i_no_slp_vectorize_ab.json) - the fallback, not the default


def _through_the_pass(hipcc, flags, include_dir, src, obj, t, log):
    """One attempt: device assembly -> rewrite -> lint -> assemble -> link -> bundle -> host object.  Returns (rewritten, lint hits)."""
    stem = obj[:-2] if obj.endswith(".o") else obj
    asm, fixed, dev_o, dev_out, fatbin = (stem + e for e in (".dev.s", ".dev.fixed.s", ".dev.o", ".dev.out", ".hipfb"))
    run = lambda cmd: subprocess.run(cmd, check=True, stdout=log, stderr=log)
    try:
        run([hipcc] + flags + ["--cuda-device-only", "-S", "-I", include_dir, src, "-o", asm])
        with open(asm) as fh:
            text, n = rewrite(fh.read())
        left = risky(text)
        if left:
            return n, left
        with open(fixed, "w") as fh:
            fh.write(text)
        run([t["clang"], "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", fixed, "-o", dev_o])
        run([t["lld"], "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_out, dev_o])
        run([t["clang-offload-bundler"], "-type=o", "-bundle-align=4096", f"-targets={OFFLOAD_TARGETS}",
             "-input=/dev/null", f"-input={dev_out}", f"-output={fatbin}"])
        run([hipcc] + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-c", "-I", include_dir, src, "-o", obj])
        return n, []
    finally:
        for f in (asm, fixed, dev_o, dev_out, fatbin):          # every intermediate; the linked libraries are what the lint test reads
            if os.path.exists(f):
                os.remove(f)


def compile_unit(hipcc, flags, include_dir, src, obj, log=None):
    """src (.hip) -> obj (host object with the device code embedded).  Returns a record {"mode", "rewritten", ...} that build() keeps
    next to the object and folds into the library's manifest.

    Modes, in the order tried (the first that yields an object without the affected form wins):
      "rewrite"                 the pass above on hipcc's assembly (no measurable cost);
      "rewrite+no-slp"          the same with -fno-slp-vectorize, when the pass met a form it cannot rewrite (an instruction outside
                                COMMUTATIVE) - the vectoriser is what emits the form, so this unit pays the flag's cost, the others do not;
      "plain+no-slp"            hipcc's own one-step compile with -fno-slp-vectorize, when the pass itself cannot run (a tool is missing,
                                or the hand-driven assemble / link / bundle steps fail under a hipcc whose driver changed): the object's
                                device code is then disassembled and linted.
    Raises only if all three leave the form in the object."""
    # hipcc derives a unit's CUID (a suffix of internal symbol names shared by its device and host halves) from the source file's
    # PATH: pin it to the unit's name so that the library's bytes do not depend on where the repository is checked out
    flags = list(flags) + ["-cuid=vmapstep_" + os.path.splitext(os.path.basename(src))[0]]
    rec = {"unit": os.path.basename(src), "flags": list(flags)}
    why = None
    try:
        t = tools(hipcc)
        n, left = _through_the_pass(hipcc, flags, include_dir, src, obj, t, log)
        if not left:
            return dict(rec, mode="rewrite", rewritten=n)
        why = f"{len(left)} affected instruction(s) the pass cannot rewrite, e.g. {left[0]}"
        n, left2 = _through_the_pass(hipcc, flags + [NO_SLP], include_dir, src, obj, t, log)
        if not left2:
            return dict(rec, mode="rewrite+no-slp", rewritten=n, fallback_reason=why)
        why += f"; with {NO_SLP} still {len(left2)}, e.g. {left2[0]}"
    except (FileNotFoundError, subprocess.CalledProcessError) as e:
        why = (why + "; " if why else "") + f"the pass could not run: {type(e).__name__}: {str(e)[:200]}"
    subprocess.run([hipcc] + flags + [NO_SLP, "-c", "-I", include_dir, src, "-o", obj], check=True, stdout=log, stderr=log)
    left = lint_object(obj)
    if left:
        raise RuntimeError(f"{src}: the affected packed-float32 form survives every build mode ({why}); in the plain {NO_SLP} object: {left[:3]}")
    return dict(rec, mode="plain+no-slp", rewritten=0, fallback_reason=why,
                linted=left is not None)


def lint_object(obj):
    """risky() over the gfx950 code embedded in one host object (llvm-objdump --offloading + -d); None if the tools to look are missing."""
    import tempfile
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    bundler = os.path.join(LLVM_BIN, "clang-offload-bundler")
    if not (os.path.exists(objdump) and os.path.exists(bundler)):
        return None
    with tempfile.TemporaryDirectory() as d:
        try:
            units = disassemble_library(obj, d)
        except subprocess.CalledProcessError:
            return None
        if not units or not any("v_" in text for _, text in units):
            return None                                           # nothing to look at is not the same as nothing found
        return [b for _, text in units for b in risky(text)]


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
