#!/usr/bin/env python3
"""kmeta.py <object|so>: per-kernel register / spill / LDS metadata of the gfx950 code objects inside a host object."""
import re, subprocess, sys, os, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
def code_objects(path):
    fat = tempfile.mktemp(suffix=".fat")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    data = open(fat, "rb").read(); os.unlink(fat)
    out = []
    # the fat binary may be compressed (CCOB) or plain (__CLANG_OFFLOAD_BUNDLE__); let the bundler handle either
    pos = 0
    idx = 0
    while True:
        i = min([p for p in (data.find(b"CCOB", pos), data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)) if p >= 0], default=-1)
        if i < 0: break
        j = min([p for p in (data.find(b"CCOB", i + 4), data.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)) if p >= 0], default=len(data))
        blob = tempfile.mktemp(suffix=".bundle"); open(blob, "wb").write(data[i:j])
        co = tempfile.mktemp(suffix=".co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={blob}", f"--output={co}"], capture_output=True)
        os.unlink(blob)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co): out.append(co)
        pos = j; idx += 1
    return out
def main():
    for path in sys.argv[1:]:
        for co in code_objects(path):
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            size = os.path.getsize(co)
            print(f"# {path}: code object {size} bytes")
            for blk in notes.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name).replace("void vk::", "")
                print(f"{name:70s} vgpr {g('vgpr_count'):>4} (spill {g('vgpr_spill_count')}) sgpr {g('sgpr_count'):>4} (spill {g('sgpr_spill_count')}) lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size')}")
            os.unlink(co)
main()
