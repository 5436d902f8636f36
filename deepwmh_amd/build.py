"""Build libdeepwmh_amd.so (hipcc, gfx950 only) in-tree so it travels with the source snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libdeepwmh_amd.so')
DIAG_LIB = os.path.join(LIBDIR, 'libdeepwmh_amd_diag.so')      # -DWMH_DIAG: cycle stamps + WMH_* variant switches (csrc/wmh_common.h)
SOURCES = ['conv3d_k3.hip', 'conv3d_k3_ws.hip', 'conv3d_k3_bf16.hip', 'conv3d_k3_bf16_ws.hip', 'conv3d_k3s2_f16x2_ws.hip', 'conv3d_k3_wgrad.hip', 'conv3d_k3_wgrad_bf16.hip', 'conv3d_k3_wgrad_f16x2.hip', 'conv3d_k3s2_dgrad.hip', 'conv3d_k3s2_dgrad_f16x2.hip', 'convT_k2s2.hip', 'convT_k2s2_f16x2.hip', 'convT_k2s2_wgrad.hip', 'train_kernels.hip', 'volume_ops.hip', 'resample.hip', 'stage1_kernels.hip', 'wmh_kernels.hip', 'cl_conv3.hip', 'cl_misc.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-result']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build_library(force=False, verbose=True, diag=True):
    """The product library and (diag=True) the diagnostic one; returns the product library's path."""
    lib = _build_one(LIB, 'obj', [], force, verbose)
    if diag:
        _build_one(DIAG_LIB, 'obj_diag', ['-DWMH_DIAG'], force, verbose)
    return lib


def _build_one(LIB, objname, extra_flags, force, verbose):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, objname)
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, 'wmh_common.h'), os.path.join(CSRC, 'conv3d_common.h'), os.path.join(CSRC, 'cl_common.h'), os.path.join(HERE, '..', 'include', 'deepwmh_amd.h')]
    objs, relink, jobs = [], force or not os.path.exists(LIB), []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            jobs.append([HIPCC] + FLAGS + extra_flags + ['-c', s, '-o', o])
            relink = True
        objs.append(o)
    # the translation units are independent: a few compilers side by side (WMH_BUILD_JOBS, default 4 -- each hipcc peaks at ~1.5 GB)
    width, running = max(1, int(os.environ.get('WMH_BUILD_JOBS', '4'))), []
    while jobs or running:
        while jobs and len(running) < width:
            cmd = jobs.pop(0)
            if verbose:
                print(' '.join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, proc = running.pop(0)
        if proc.wait() != 0:
            for _, other in running:
                other.kill()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if relink:
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
    print(LIB)
