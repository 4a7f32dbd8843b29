#!/usr/bin/env python3
"""List scratch (spill) accesses and loop labels of one kernel in the saved gfx950 assembly (make asm).
usage: tools/asm_scan.py [substring of the mangled name, default ILi32ELi0ELb0E]"""
import re, sys
key = sys.argv[1] if len(sys.argv) > 1 else "ILi32ELi0ELb0E"
s = open('/tmp/tsdf_hip-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\w*tsdf_fused_kernel' + key + r'\w*:', l))
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
body = s[start:end]
print(len(body), "lines")
for k, l in enumerate(body):
    t = l.strip()
    if 'scratch_' in t or 's_barrier' in t or 'global_load_lds' in t:
        print(k, t)
