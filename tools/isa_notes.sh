#!/bin/bash
# Register / scratch / LDS use of every kernel in one object of libl2a_hip.so (developer aid):
#   bash tools/isa_notes.sh learning_to_adapt_amd/csrc/_obj/l2a_api.o [name filter]
# object -> .hip_fatbin section -> clang-offload-bundler -> llvm-readelf --notes
OBJ=$1; FILT=${2:-.}
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin $OBJ $T/fat.bin
tgt=$($LLVM/clang-offload-bundler --list --type=o --input=$T/fat.bin | grep gfx950 | head -1)
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=$tgt --output=$T/dev.co
$LLVM/llvm-readelf --notes $T/dev.co | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count" \
  | awk '/\.name:/ {if (n) print n, rest; n=$2; rest=""} !/\.name:/ {rest = rest " " $1 $2} END {print n, rest}' | grep -E "$FILT" | sed 's/\.//g'
rm -rf $T
