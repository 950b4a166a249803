#!/bin/bash
# usage: tools/kernel_meta.sh <object or .so> [name filter]: VGPRs / SGPRs / spills / LDS / scratch of every gfx950 kernel in it
set -e
d=$(mktemp -d)
cp "$1" "$d/in.o"
(cd "$d" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.o >/dev/null 2>&1 || true)
for f in "$d"/*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | grep -E "^ +\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|agpr_count):" \
    | awk '/\.agpr_count/{a=$2} /\.group_segment/{g=$2} /\.name:/{n=$2} /\.private_segment/{p=$2} /\.sgpr_count/{s=$2} /\.sgpr_spill/{ss=$2} /\.vgpr_count/{v=$2} /\.vgpr_spill/{vs=$2; printf "%s vgpr=%s agpr=%s sgpr=%s vspill=%s sspill=%s scratch=%s lds=%s\n", n, v, a, s, vs, ss, p, g}'
done | grep -E "${2:-.}" || true
/bin/rm -r "$d"
