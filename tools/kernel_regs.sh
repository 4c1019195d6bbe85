#!/bin/bash
# registers / scratch / LDS of the device kernels of one object: tools/kernel_regs.sh [object] [name regex]
obj=$(realpath ${1:-loam_amd/lib/obj/register_kernels.o}); re=${2:-.}
tmp=$(mktemp -d); cd "$tmp"
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin "$obj" x.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev.co | c++filt | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(.+)', blk) or [None,'?'])[1].strip()
    name=g('name')
    if re.search(sys.argv[1], name): print('%-100s vgpr %3s agpr %s sgpr %3s scratch %4s lds %6s'%(name[:100],g('vgpr_count'),blk.split()[1] if blk.strip().startswith(':') else '?',g('sgpr_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
" "$re"
rm -rf "$tmp"
