#!/bin/bash
# tools/build_sed_flags_variant.sh NAME 'sed-expression' FILE 'extra hipcc flags' -- experiment aid: gpurun_in/lib_NAME.so from csrc with the sed expression
# applied to csrc/FILE and extra compiler flags (tools/build_variant.sh + tools/build_flags_variant.sh in one); the product library is not touched
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/svt-vp9_amd $T/include $ROOT/gpurun_in
cp -r $ROOT/svt-vp9_amd/csrc $ROOT/svt-vp9_amd/host $ROOT/svt-vp9_amd/shim $ROOT/svt-vp9_amd/Makefile $T/svt-vp9_amd/
cp $ROOT/include/*.h $T/include/
sed -i "$2" $T/svt-vp9_amd/csrc/$3
BASE=$(grep '^HIPFLAGS' $ROOT/svt-vp9_amd/Makefile | sed 's/^HIPFLAGS := //; s/\$(ARCH)/gfx950/')
(cd $T/svt-vp9_amd && rm -f csrc/*.o host/*.o && make -s -j8 libsvtvp9_hip.so HIPFLAGS="$BASE $4" 2>&1 | grep -E "error" || true)
cp $T/svt-vp9_amd/libsvtvp9_hip.so $ROOT/gpurun_in/lib_$1.so
rm -rf $T
