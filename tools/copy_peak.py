"""Measured HBM copy rate (kb_measure_copy_bandwidth) for a few sizes."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kbmod_amd import capi
lib = capi.load_lib()
for mb in (256, 1024, 4096):
    g = C.c_double()
    capi.check(lib.kb_measure_copy_bandwidth(mb << 20, 10, None, C.byref(g)))
    print(f"{mb} MiB each way: {g.value:.0f} GB/s (read + write)")
for mb in (32, 128, 200, 512, 4096):
    g = C.c_double()
    capi.check(lib.kb_measure_read_bandwidth(mb << 20, 20, None, C.byref(g)))
    print(f"{mb} MiB read-only: {g.value:.0f} GB/s")
