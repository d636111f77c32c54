"""Measurement builds: tests/tools/build_variant.py <tag> <unit>[,<unit>...] -DFLAG[=V] ... -> tests/tools/libvmapstep_<tag>.so =
the product library with the named translation units recompiled with the extra flags (everything else linked as the product
built it).  Run a tool / bench.py on it with VMAPSTEP_LIBRARY=tests/tools/libvmapstep_<tag>.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    tag, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(ROOT, "tests", "tools", f"_var_{tag}")
    os.makedirs(objdir, exist_ok=True)
    procs = [subprocess.Popen([hipcc] + ge.HIPCC_FLAGS + flags + ["-c", "-I", ge.CSRC, os.path.join(ge.CSRC, u + ".hip"), "-o", os.path.join(objdir, u + ".o")])
             for u in units]
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("hipcc failed")
    all_units = [f[:-4] for f in sorted(os.listdir(ge.CSRC)) if f.endswith(".hip")]
    objs = [os.path.join(objdir if u in units else os.path.join(ge.CSRC, "_obj"), u + ".o") for u in all_units]
    out = os.path.join(ROOT, "tests", "tools", f"libvmapstep_{tag}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
    print(out)


if __name__ == "__main__":
    main()
