"""Builds the REFERENCE's own TSDF-fusion kernel as a checker (test infrastructure; nothing in the product path loads it).

The reference's only native code is the CUDA C string it hands to PyCUDA's SourceModule (/root/reference/src/fusion.py:69-142).
pycuda / numba are absent from this image, so the reference's `TSDFVolume` cannot run -- but the kernel string is plain CUDA C
that hipcc compiles as it stands.  This script

  * reads the string out of /root/reference/src/fusion.py WHERE IT LIES (nothing of it is copied into the repository: the
    extracted text lives in a temporary directory for the duration of the compile),
  * appends a launcher of OUR OWN (`ref_fusion_integrate`: what PyCUDA's `self._cuda_integrate(...)` call at :226-251 does -- device
    pointers in, one launch per `gpu_loop_idx` with the grid the caller computed as :146-154 does),
  * compiles it for gfx950 into oracle/_ref/ (git-ignored, travels to the GPU box with the snapshot) -- twice:
        libref_fusion_exact.so     -ffp-contract=off : every operation rounded as written (what the numpy restatement
                                   oracle.tsdf_integrate_np and the product kernel csrc/adfp_fusion.h reproduce bit for bit)
        libref_fusion_contract.so  hipcc's default contraction (mul + add -> fma where the compiler chooses, like nvcc's default
                                   -fmad=true under PyCUDA): the compiler-dependent variant, compared at float32 rounding.

tests/test_gpu_fusion.py runs both against adfp_tsdf_integrate on the MI355X.  `python oracle/build_ref_fusion.py` (also called by
__graft_entry__.build() when /root/reference is present)."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src/fusion.py'
OUT = os.path.join(HERE, '_ref')

LAUNCHER = r'''
// ---- launcher (ours): the PyCUDA call of src/fusion.py:226-251 with device pointers
extern "C" int ref_fusion_integrate(float* tsdf_vol, float* weight_vol, float* color_vol, float* vol_dim, float* vol_origin, float* cam_intr,
                                    float* cam_pose, float* other_params /* device, [n_loops][6] */, float* color_im, float* depth_im,
                                    int n_loops, int grid_x, int grid_y, int grid_z, int block, void* stream) {
    for (int k = 0; k < n_loops; ++k) {
        hipLaunchKernelGGL(integrate, dim3(grid_x, grid_y, grid_z), dim3(block, 1, 1), 0, (hipStream_t)stream, tsdf_vol, weight_vol, color_vol,
                           vol_dim, vol_origin, cam_intr, cam_pose, other_params + 6 * k, color_im, depth_im);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
'''


def kernel_string():
    src = open(REF).read()
    m = re.search(r'SourceModule\("""(.*?)"""\)', src, re.S)
    if not m:
        raise RuntimeError('the SourceModule string was not found in ' + REF)
    return m.group(1)


def build(hipcc='/opt/rocm/bin/hipcc'):
    if not os.path.exists(REF):
        return False
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'ref_integrate.hip')
        with open(path, 'w') as f:
            f.write('#include <hip/hip_runtime.h>\n' + kernel_string() + '\n' + LAUNCHER)
        for name, flags in (('libref_fusion_exact.so', ['-ffp-contract=off']), ('libref_fusion_contract.so', [])):
            subprocess.check_call([hipcc, '-O3', '--offload-arch=gfx950', '-shared', '-fPIC'] + flags + ['-o', os.path.join(OUT, name), path])
    return True


if __name__ == '__main__':
    ok = build()
    print('built' if ok else f'{REF} is not present: nothing built', file=sys.stderr)
