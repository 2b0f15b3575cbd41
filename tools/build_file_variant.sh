#!/bin/bash
# tools/build_file_variant.sh NAME FILE_UNDER_CSRC [GIT_REV] -- experiment aid: builds gpurun_in/lib_NAME.so from the working tree with
# csrc/FILE taken from GIT_REV (default HEAD): an A/B partner for a kernel that is being changed.  The product library is not
# touched.  Run against it with SVT_HIP_LIB=gpurun_in/lib_NAME.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/svt-vp9_amd $T/include $ROOT/gpurun_in
cp -r $ROOT/svt-vp9_amd/csrc $ROOT/svt-vp9_amd/host $ROOT/svt-vp9_amd/shim $ROOT/svt-vp9_amd/Makefile $T/svt-vp9_amd/
cp $ROOT/include/*.h $T/include/
(cd $ROOT && git show ${3:-HEAD}:svt-vp9_amd/csrc/$2) > $T/svt-vp9_amd/csrc/$2
(cd $T/svt-vp9_amd && rm -f csrc/*.o host/*.o && make -s -j8 libsvtvp9_hip.so 2>&1 | grep -E "error" || true)
cp $T/svt-vp9_amd/libsvtvp9_hip.so $ROOT/gpurun_in/lib_$1.so
rm -rf $T
