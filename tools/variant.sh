#!/bin/bash
# Build a variant of libresnet_mi.so with extra compiler defines, for A/B runs in one gpurun call:
#   tools/variant.sh NAME "-DPC_ABLATE=1" [file.hip ...]   ->  variants/libresnet_mi_NAME.so   (use with RESNET_MI_LIB=...)
# Only the listed .hip files are recompiled with the defines (default: all of them); objects of the normal build are reused.
set -e
NAME=$1; DEFS=$2; shift 2 || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/resnet_amd/csrc
OUT=$ROOT/variants; mkdir -p "$OUT/obj_$NAME"
make -s -C "$SRC"
FILES=${@:-$(cd $SRC && ls *.hip)}
OBJS=""
for f in $(cd $SRC && ls *.hip *.c); do
  o=$SRC/${f%.*}.o
  for v in $FILES; do
    if [ "$v" = "$f" ]; then
      o=$OUT/obj_$NAME/${f%.*}.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -I$SRC -I$ROOT/include $DEFS -c $SRC/$f -o $o
    fi
  done
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libresnet_mi_$NAME.so $OBJS -ldl -lm
echo "$OUT/libresnet_mi_$NAME.so"
