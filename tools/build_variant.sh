#!/bin/bash
# Build an alternative gkmkern_pylib.so for A/B timing (tools/kernel_ab.py loads it through GKM_LIB_PATH):
#   tools/build_variant.sh <name> ["<extra hipcc flags>"] [<git revision of gkmqc_amd/csrc to build instead of the working tree>]
# -> build_variants/lib_<name>.so   (git-ignored; travels to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=${2:-}; REV=${3:-}
OUT=$PWD/build_variants
mkdir -p "$OUT/obj_$NAME"
SRC=$PWD/gkmqc_amd/csrc
if [ -n "$REV" ]; then
  SRC=$OUT/src_$NAME/gkmqc_amd/csrc
  rm -rf "$OUT/src_$NAME"; mkdir -p "$SRC" "$OUT/src_$NAME/include"
  for f in $(git ls-tree --name-only "$REV" gkmqc_amd/csrc/ include/); do git show "$REV:$f" > "$OUT/src_$NAME/$f"; done
fi
make -s -C "$SRC" BUILD="$OUT/obj_$NAME" BIN="$OUT/bin_$NAME" EXTRA="$EXTRA" "$OUT/bin_$NAME/gkmkern_pylib.so"
cp "$OUT/bin_$NAME/gkmkern_pylib.so" "$OUT/lib_$NAME.so"
echo "built $OUT/lib_$NAME.so"
