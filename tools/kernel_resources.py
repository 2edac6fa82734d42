#!/usr/bin/env python3
"""Register / spill / scratch figures of every kernel of libkbmod_hip.so (from the code-object notes).

    python tools/kernel_resources.py [substring ...]
    KB_OBJ_DIR=/tmp/objs python tools/kernel_resources.py ...   (objects of a variant build)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.environ.get("KB_OBJ_DIR", os.path.join(ROOT, "kbmod_amd", "_obj"))
LLVM = "/opt/rocm/lib/llvm/bin"


def notes_of(obj, tmp):
    # the device code object sits in the .hip_fatbin section of each translation unit's object as a clang offload bundle
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    if os.path.getsize(fat) == 0:
        return ""  # no device code in this translation unit
    co = os.path.join(tmp, "gfx950.co")
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
    return subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout


def main():
    want = sys.argv[1:]
    kernels = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(os.listdir(OBJ)):
            if not obj.endswith(".o"):
                continue
            cur = None
            for line in notes_of(os.path.join(OBJ, obj), tmp).split("\n"):
                if line.startswith("  - .agpr_count"):
                    cur = {}
                    kernels.append(cur)
                m = re.match(r"\s+-?\s*\.(name|sgpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|"
                             r"private_segment_fixed_size|group_segment_fixed_size):\s+(.*)", line)
                if m and cur is not None and line.startswith("    ."):
                    cur[m.group(1)] = m.group(2)
    kernels = [k for k in kernels if "name" in k and "vgpr_count" in k]
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in kernels), capture_output=True,
                           text=True).stdout.split("\n")
    print(f"{'kernel':86s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
    for k, n in zip(kernels, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if want and not any(w in n for w in want):
            continue
        print(f"{n[:86]:86s} {k['vgpr_count']:>5s} {k['sgpr_count']:>5s} {k.get('vgpr_spill_count', '0'):>6s} "
              f"{k.get('sgpr_spill_count', '0'):>6s} {k.get('private_segment_fixed_size', '0'):>7s} "
              f"{k.get('group_segment_fixed_size', '0'):>6s}")


if __name__ == "__main__":
    main()
