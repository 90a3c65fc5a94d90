import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
L = mas._lib.lib()
for mode in (0, 1, 2):
    for n in (50, 200):
        us = C.c_double()
        st = L.mis_debug_launch_floor(0, n, mode, 20, C.byref(us))
        print("mode", mode, "n", n, "status", st, "us/kernel %.2f" % us.value, flush=True)
