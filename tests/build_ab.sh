#!/bin/bash
# Build the HIP library of a git revision (or of the working tree: REV = WORK) into build/ab4/lib_NAME.so, for A/B runs on the GPU box
# (tests/gpu_ab.py build/ab4/lib_*.so).  Usage: bash tests/build_ab.sh REV NAME
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/ab4; mkdir -p $OUT
TMP=$(mktemp -d)
if [ "$REV" = WORK ]; then cp -r $ROOT/boundmpc_amd $ROOT/include $TMP/; else git -C $ROOT archive $REV boundmpc_amd include | tar -x -C $TMP; fi
rm -f $TMP/boundmpc_amd/csrc/*.so
(cd $TMP && python -c "
import sys; sys.path.insert(0, '.')
from boundmpc_amd import build
print(build.build(force=True))")
cp $TMP/boundmpc_amd/csrc/libboundmpc_hip.so $OUT/lib_$NAME.so
rm -rf $TMP
echo $OUT/lib_$NAME.so
