"""In-tree build of the two native artefacts (no JIT cache, no pip install):

* ``kbmod_amd/lib/libkbmod_hip.so`` -- the C-ABI device library (hipcc, gfx950)
* ``kbmod_amd/search.<abi>.so``     -- the pybind11 host layer (g++), linked
  against the device library with an ``$ORIGIN`` rpath.

``hipcc`` cross-compiles gfx950 without a GPU, so this runs in the CPU-only
container; the built files travel to the GPU box with the repository snapshot.
"""

import os
import subprocess
import sys
import sysconfig

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")
_INC = os.path.join(_ROOT, "include")
_LIBDIR = os.path.join(_PKG, "lib")

HIP_SOURCES = ["search_lds.hip", "search_lds_encoded.hip", "search_direct.hip", "search_kernels.hip", "sigmag_kernels.hip",
               "result_kernels.hip", "exchange_kernels.hip", "device_memory.hip", "image_kernels.hip", "stamp_kernels.hip", "fits_kernels.hip"]
HIP_HEADERS = ["kb_common.h", "search_math.h", "search_common.h", "search_device.h", "search_lds.h", "search_lds_asm.h", "wave_ops.h"]
HOST_SOURCES = ["host/bindings.cpp"]
HOST_HEADERS = ["host/common.h", "host/image_utils.h", "host/psi_phi_array.h", "host/trajectory_list.h",
                "host/stack_search.h", "host/device_stack.h"]


def hip_lib_path():
    return os.path.join(_LIBDIR, "libkbmod_hip.so")


def host_module_path():
    return os.path.join(_PKG, "search" + sysconfig.get_config_var("EXT_SUFFIX"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.sep not in c or os.path.exists(c):
            return c
    return "hipcc"


def build_hip(force=False, verbose=False):
    """Compile every .hip translation unit to an object (in parallel) and link the C-ABI library."""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(_LIBDIR, exist_ok=True)
    objdir = os.path.join(_PKG, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(_CSRC, h) for h in HIP_HEADERS] + [os.path.join(_INC, "kbmod_hip.h")]
    flags = [
        "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
        # the reference CPU build has no FMA; keep every multiply and add separately rounded
        "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
        "-I" + _INC, "-I" + _CSRC,
    ]

    def compile_one(name):
        src = os.path.join(_CSRC, name)
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return obj, True
        if verbose:
            print("kept (up to date):", os.path.relpath(obj, _ROOT))
        return obj, False

    with ThreadPoolExecutor(max_workers=6) as pool:
        results = list(pool.map(compile_one, HIP_SOURCES))
    objs = [o for o, _ in results]
    out = hip_lib_path()
    if force or any(changed for _, changed in results) or not os.path.exists(out):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_host(force=False, verbose=False):
    import pybind11

    srcs = [os.path.join(_CSRC, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(_CSRC, h) for h in HOST_HEADERS] + [os.path.join(_INC, "kbmod_hip.h"), hip_lib_path()]
    out = host_module_path()
    if not force and not _stale(out, deps):
        if verbose:
            print("kept (up to date):", os.path.relpath(out, _ROOT))
        return out
    cmd = [
        "g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fvisibility=hidden",
        "-I" + _INC, "-I" + os.path.join(_CSRC, "host"), "-I" + pybind11.get_include(),
        "-I" + sysconfig.get_paths()["include"],
    ] + srcs + ["-L" + _LIBDIR, "-lkbmod_hip", "-Wl,-rpath,$ORIGIN/lib", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
