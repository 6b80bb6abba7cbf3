#!/usr/bin/env python3
"""Per step_kernel instantiation of a device-only assembly dump (hipcc ... --cuda-device-only -S): instruction lines, SGPR spills into
VGPR lanes (v_writelane / v_readlane), scratch accesses, vector loads / stores and s_waitcnt vmcnt sites.

    hipcc <Makefile FLAGS minus -shared -fPIC> --cuda-device-only -S -o /tmp/q.s gym_rotor_amd/csrc/quadrotor_kernels.hip
    python tools/isa_stats.py /tmp/q.s [substring of the mangled name]
"""
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n(?=_ZN2qr11step_kernel\S*:)', txt):
    m = re.match(r'(_ZN2qr11step_kernel\S+):', f)
    if not m or want not in m.group(1):
        continue
    body = f.split('.Lfunc_end')[0]
    insts = [l for l in body.splitlines() if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    tag = re.search(r'ILi(\d)E(\w)(\w)Li64EL[bi](\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)E', m.group(1))
    name = "kind=%s %s%s TRAJ=%s ADAPT=%s POLICY=%s SINGLE=%s HELP=%s" % tag.groups() if tag else m.group(1)
    print(name, '| insts', len(insts), 'writelane', body.count('v_writelane'), 'readlane', body.count('v_readlane'),
          'scratch', body.count('scratch_'), 'vload', len(re.findall(r'(buffer|global)_load', body)),
          'vstore', len(re.findall(r'(buffer|global)_store', body)), 'waitcnt_vm', len(re.findall(r's_waitcnt.*vmcnt', body)))
