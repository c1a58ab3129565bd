"""Debug build only (-DADFP_STAMPS): wave-cycles per phase of k_decode_bwd_fused over N fused Mapper iterations.
  hipcc ... -DADFP_STAMPS -o build/libadfp_stamps.so ; ADFP_LIB_PATH=$PWD/build/libadfp_stamps.so python tools/fused_phases.py"""
import ctypes as C
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attentive_dfprior_amd import _lib  # noqa: E402

L = _lib.lib()
L.adfp_debug_phases_fused.argtypes = [C.c_void_p, C.c_int]
ph = (C.c_ulonglong * 8)()
sys.argv = ['profile_iteration.py', '--rays', '5000', '--samples', '48', '--masked', '--iters', '20']
import torch  # noqa: E402
torch.cuda.init()
L.adfp_debug_phases_fused(ph, 1)
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profile_iteration.py'), run_name='__main__')
L.adfp_debug_phases_fused(ph, 1)
names = ['loop', 'point/masks/gout', 'wait DMA', 'c + output layer', 'five layers', 'Fourier blocks', '-', '-']
tot = float(sum(ph[:6]))
tiles = 23 * (5000 * 64 / 32)          # 3 warm-up + 20 timed iterations of the colour decoder
print({n: (round(ph[k] / tot, 3), round(ph[k] / tiles)) for k, n in enumerate(names[:6])}, '(share, clock64 ticks per tile)')
