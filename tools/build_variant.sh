#!/bin/bash
# tools/build_variant.sh NAME 'sed-expression' [file under csrc, default me_core.h] -- experiment aid: builds gpurun_in/lib_NAME.so from csrc with the sed expression
# applied to me_core.h (the product library is not touched).  Run a bench against it with SVT_HIP_LIB=gpurun_in/lib_NAME.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/svt-vp9_amd $T/include
cp -r $ROOT/svt-vp9_amd/csrc $ROOT/svt-vp9_amd/host $ROOT/svt-vp9_amd/shim $ROOT/svt-vp9_amd/Makefile $T/svt-vp9_amd/
cp $ROOT/include/*.h $T/include/
sed -i "$2" $T/svt-vp9_amd/csrc/${3:-me_core.h}
(cd $T/svt-vp9_amd && rm -f csrc/*.o host/*.o && make -s libsvtvp9_hip.so 2>&1 | grep -E "error" || true)
cp $T/svt-vp9_amd/libsvtvp9_hip.so $ROOT/gpurun_in/lib_$1.so
rm -rf $T
