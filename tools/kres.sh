#!/bin/bash
# kernel resource usage (VGPRs, spills, LDS, scratch) of the gfx950 code object inside a HIP object file / .so
#   tools/kres.sh gwen_amd/build/wide.o [name filter]
set -e
f=$1; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$tmp/fat.bin" "$f"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat.bin" --output="$tmp/dev.co" --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/dev.co" | awk '
/\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2} /\.sgpr_count:/ {s=$2}
/\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2} /\.agpr_count:/ {a=$2}
/\.wavefront_size:/ {print name, "vgpr="v, "sgpr="s, "spill="sp, "scratch="p, "lds="l}' | grep -E "$pat" | while read n rest; do echo "$(echo $n | c++filt) $rest"; done
rm -rf "$tmp"
