#!/bin/bash
# The library as it was at an older commit, for same-box A/B runs (tools/ab_libs.sh, tools/pmc_layer.py): sources of <commit> -> ab_old/<tag>/
# (git-ignored; travels to the GPU box).  Usage (build container): bash tools/build_old_lib.sh <commit> <tag> [extra hipcc flags, e.g. -DWMH_DIAG]
set -e
commit=$1; tag=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/ab_old/$tag
rm -rf $dst; mkdir -p $dst/src/deepwmh_amd $dst/obj
git -C $root archive $commit deepwmh_amd/csrc include | tar -x -C $dst/src
cd $dst/src/deepwmh_amd/csrc
ls *.hip | xargs -P 6 -I{} sh -c "/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result $* -c {} -o $dst/obj/{}.o 2>/dev/null"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $dst/libdeepwmh_amd.so $dst/obj/*.o
rm -rf $dst/obj $dst/src
ls -la $dst
