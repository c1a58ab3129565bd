"""Regenerates profiles/r06_isa_mix_lc16.txt: the instruction mix of k_decode_lc16<768>'s TILE LOOP (and of the whole kernel beside
it) from the CURRENT sources, stamped with bench.source_hash() -- the input of bench.py's `roofline.limiter` model
(tests/test_host_logic.py checks the stamp against the sources, so a kernel change without a regeneration fails the CPU suite).

  python tools/gen_isa_mix.py            (hipcc -S with csrc/build.sh's flags into a temporary directory, ~40 s; no GPU needed)
"""
import io
import os
import subprocess
import sys
import tempfile
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench                                                         # noqa: E402
import isa_mix                                                       # noqa: E402

OUT = os.path.join(ROOT, 'profiles', 'r06_isa_mix_lc16.txt')
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize']


def main():
    src = os.path.join(ROOT, 'attentive_dfprior_amd', 'csrc', 'adfp_kernels.hip')
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 'adfp.s')
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + FLAGS + ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only',
                                                                                            '-o', asm, src])
        buf = io.StringIO()
        with redirect_stdout(buf):
            for loop in (True, False):
                sys.argv = ['isa_mix.py'] + (['--loop'] if loop else []) + [asm, 'k_decode_lc16', 'ILi768E']
                isa_mix.main()
    with open(OUT, 'w') as f:
        f.write(f'# source_hash={bench.source_hash()}  tools/gen_isa_mix.py: hipcc {" ".join(FLAGS)} -S --cuda-device-only; tools/isa_mix.py [--loop] k_decode_lc16 ILi768E\n')
        f.write('# first entry: the TILE LOOP only (the smallest loop holding every MFMA: one 32-point tile, both networks); second: the whole kernel\n')
        f.write(buf.getvalue())
    print(open(OUT).read()[:1500])


if __name__ == '__main__':
    main()
