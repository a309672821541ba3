#!/bin/bash
# compact instruction-class stream of one kernel of a HIP object (M mfma, v VALU, s SALU, R/W ds read/write, D lds-dma,
# S/G global store/load, L scalar load, [..] s_waitcnt, j branch, n s_nop):   tools/kstream.sh build/wide.o "256, 256, 8, 1, 128, true, false, 2"
set -e
f=$1; pat=$2; tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$tmp/fat.bin" "$f"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat.bin" --output="$tmp/dev.co" --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$tmp/dev.co" | c++filt > "$tmp/all.s"
python3 - "$tmp/all.s" "$pat" <<'PY'
import re, sys
txt = open(sys.argv[1]).read().split('\n')
on = False; out = []
def cls(op):
    for pre, c in (('v_mfma', 'M'), ('ds_read', 'R'), ('ds_load', 'R'), ('ds_write', 'W'), ('ds_store', 'W'), ('global_load_lds', 'D'),
                   ('global_store', 'S'), ('global_load', 'G'), ('scratch_', 'X'), ('s_load', 'L'), ('s_buffer', 'L'), ('s_cbranch', 'j'),
                   ('s_branch', 'j'), ('s_nop', 'n'), ('s_', 's'), ('v_', 'v')):
        if op.startswith(pre): return c
    return '?'
for l in txt:
    if re.match(r'^[0-9a-f]+ <', l):
        on = sys.argv[2] in l
        continue
    if not on: continue
    m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', l)
    if not m: continue
    op, args = m.group(1), m.group(2)
    if op.startswith('s_waitcnt'): out.append('[' + args.replace('vmcnt', 'vm').replace('lgkmcnt', 'lg').replace(' ', '') + ']')
    elif op.startswith('s_barrier'): out.append('\n|BARRIER|\n')
    else: out.append(cls(op))
print(''.join(out))
PY
rm -rf "$tmp"
