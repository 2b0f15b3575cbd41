#!/bin/bash
# tools/build_flags_variant.sh NAME 'extra hipcc flags' -- experiment aid: builds gpurun_in/lib_NAME.so with extra compiler flags
# (the product library is not touched).  Run a bench against it with SVT_HIP_LIB=gpurun_in/lib_NAME.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/svt-vp9_amd $T/include
cp -r $ROOT/svt-vp9_amd/csrc $ROOT/svt-vp9_amd/host $ROOT/svt-vp9_amd/shim $ROOT/svt-vp9_amd/Makefile $T/svt-vp9_amd/
cp $ROOT/include/*.h $T/include/
BASE=$(grep '^HIPFLAGS' $ROOT/svt-vp9_amd/Makefile | sed 's/^HIPFLAGS := //; s/\$(ARCH)/gfx950/')
(cd $T/svt-vp9_amd && rm -f csrc/*.o host/*.o && make -s -j8 libsvtvp9_hip.so HIPFLAGS="$BASE $2" 2>&1 | grep -E "error" || true)
cp $T/svt-vp9_amd/libsvtvp9_hip.so $ROOT/gpurun_in/lib_$1.so
rm -rf $T
