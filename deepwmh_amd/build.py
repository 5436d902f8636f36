"""Build libdeepwmh_amd.so (hipcc, gfx950 only) in-tree so it travels with the source snapshot.

Every translation unit is compiled with -Rpass-analysis=kernel-resource-usage; the remarks (registers, scratch, LDS per kernel) are kept next
to the object and collected into lib/<library>.resources.json, which tests/test_build_resources.py reads: a hot kernel that starts spilling
-- as the strided pair kernel did when round 4 compiled its cycle stamps out (1.6x slower, nothing else changed) -- fails the CPU suite."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libdeepwmh_amd.so')
DIAG_LIB = os.path.join(LIBDIR, 'libdeepwmh_amd_diag.so')      # -DWMH_DIAG: cycle stamps + WMH_* variant switches (csrc/wmh_common.h)
SOURCES = ['conv3d_k3.hip', 'conv3d_k3_ws.hip', 'conv3d_k3_bf16.hip', 'conv3d_k3_bf16_ws.hip', 'conv3d_k3s2_f16x2_ws.hip', 'conv3d_up_f16x2.hip', 'conv3d_k3_wgrad.hip', 'conv3d_k3_wgrad_f16x2.hip', 'conv3d_k3s2_dgrad.hip', 'conv3d_k3s2_dgrad_f16x2.hip', 'convT_k2s2.hip', 'convT_k2s2_f16x2.hip', 'convT_k2s2_wgrad.hip', 'train_kernels.hip', 'volume_ops.hip', 'resample.hip', 'stage1_kernels.hip', 'wmh_kernels.hip', 'cl_conv3.hip', 'cl_misc.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-result', '-Rpass-analysis=kernel-resource-usage']


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
        # (an object without its .res file -- the resource remarks -- is recompiled: the JSON below is rebuilt from those files)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps) or not os.path.exists(o[:-2] + '.res'):
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
            running.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        cmd, proc = running.pop(0)
        _, err = proc.communicate()
        remarks = [l for l in err.splitlines() if 'kernel-resource-usage' in l]
        rest = [l for l in err.splitlines() if 'kernel-resource-usage' not in l]
        if rest and (verbose or proc.returncode != 0):
            print('\n'.join(rest), file=sys.stderr, flush=True)
        if proc.returncode != 0:
            for _, other in running:
                other.kill()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        with open(cmd[-1][:-2] + '.res', 'w') as f:         # <object>.res: the unit's resource remarks
            f.write('\n'.join(remarks) + '\n')
    res = {}
    for o in objs:
        unit, cur = os.path.basename(o)[:-2], None
        try:
            lines = open(o[:-2] + '.res').read().splitlines()
        except OSError:
            continue
        for l in lines:
            m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', l)
            if not m:
                continue
            t = m.group(1)
            if t.startswith('Function Name:'):
                cur = res.setdefault(t.split(':', 1)[1].strip(), dict(unit=unit))
            elif cur is not None and ':' in t:
                k, v = t.split(':', 1)
                key = {'VGPRs': 'vgprs', 'AGPRs': 'agprs', 'TotalSGPRs': 'sgprs', 'ScratchSize [bytes/lane]': 'scratch_bytes_per_lane',
                       'LDS Size [bytes/block]': 'static_lds_bytes', 'Occupancy [waves/SIMD]': 'waves_per_simd'}.get(k.strip())
                if key:
                    cur[key] = int(v.strip())
    with open(LIB[:-3] + '.resources.json', 'w') as f:
        json.dump(res, f, indent=0, sort_keys=True)
    if relink:
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
    print(LIB)
