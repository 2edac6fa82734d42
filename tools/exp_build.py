"""Times kb_build_psi_phi_from_device on a T x N x N stack with HIP events (builder timing experiments)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kbmod_amd import capi, fake_data as fd
T, N = int(sys.argv[1]), int(sys.argv[2])
FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # KB_BUILD_SEPARABLE = 1, KB_BUILD_GENERAL_TILES = 4
NB = int(sys.argv[4]) if len(sys.argv) > 4 else -1
lib = capi.load_lib()
sci = torch.randn((T, N, N), device="cuda"); var = torch.full((T, N, N), 4.0, device="cuda")
psf = fd.make_gaussian_kernel(1.0)
psf_all = np.ascontiguousarray(np.tile(psf.ravel(), T), dtype=np.float32); dims = np.full(T, psf.shape[0], dtype=np.int32)
stream = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    meta, arr = capi.Meta(), C.c_void_p()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    capi.check(lib.kb_build_psi_phi_from_device_ex(sci.data_ptr(), var.data_ptr(), psf_all.ctypes.data, dims.ctypes.data, T, N, N, NB, FLAGS, C.byref(meta), C.byref(arr), stream))
    e1.record(); torch.cuda.synchronize()
    lib.kb_free_gpu_block(arr)
print(sys.argv[1:], "build device ms", round(e0.elapsed_time(e1), 3))
