#!/bin/bash
# Usage: tools/isa_summary.sh <file.hip> [kernel-name-substring]
# Compiles one source for gfx950 with -save-temps and prints, per kernel, VGPRs / occupancy and the compressed
# sequence of memory instructions, waits, barriers and MFMAs (L = global load, S = store, w = vmcnt wait, B = barrier,
# M = mfma, | = basic-block boundary; runs are counted): the quick way to see whether the loads of a loop leave
# back to back ("8L w") or are serialised by waits and branches ("L w | L w | ...").
set -e
src=$(realpath "$1"); pat="${2:-}"
tmp=/tmp/isa_$(basename "$src" .hip); mkdir -p "$tmp"; cd "$tmp"
extra=""
case "$(basename "$src")" in multibox.hip|detect.hip|optim.hip) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra -c "$src" -o out.o -save-temps -Rpass-analysis=kernel-resource-usage 2> remarks.txt || { cat remarks.txt; exit 1; }
asm=$(ls *-hip-amdgcn-amd-amdhsa-gfx950.s)
grep -n "^_Z.*:.*@" "$asm" | while IFS=: read -r line name rest; do
  dem=$(echo "$name" | c++filt | sed -E 's/\(anonymous namespace\):://g; s/\(.*//' | cut -c1-110)
  echo "$dem" | grep -qi -- "$pat" || continue
  echo "=== $dem"
  grep -A9 "Function Name: $name" remarks.txt | grep -E "VGPRs:|Occupancy|ScratchSize|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste -sd' '
  sed -n "$line,\$p" "$asm" | awk '/^\.Lfunc_end/{exit} {print}' | grep -E "global_load|global_store|global_atomic|buffer_|s_waitcnt vmcnt|s_barrier|v_mfma|^\.LBB|scratch_" | sed -E 's/^\s+//' \
    | awk '{ if ($1=="s_waitcnt") print "w"; else if ($1 ~ /^\.LBB/) print "|"; else if ($1 ~ /load/) print "L"; else if ($1 ~ /store|atomic/) print "S"; else if ($1=="s_barrier") print "B"; else if ($1 ~ /mfma/) print "M"; else print $1}' \
    | uniq -c | awk '{printf "%s%s ", ($1>1?$1:""), $2} END {print ""}' | sed -E 's/(\| )+/| /g' | fold -w 220
done
