#!/usr/bin/env python3
"""Per step_kernel instantiation of a device-only assembly dump (hipcc ... --cuda-device-only -S): instruction lines, SGPR spills into
VGPR lanes (v_writelane / v_readlane), scratch accesses, vector loads / stores and s_waitcnt vmcnt sites — for the whole kernel
and, for the multi-step instantiations, INSIDE the per-env-step loop (the outermost loop of the stepping wave = the widest span
between a label and a backward branch to it): that is where an instruction is paid once per env-step.

    hipcc <Makefile FLAGS minus -shared -fPIC> --cuda-device-only -S -o /tmp/q.s gym_rotor_amd/csrc/quadrotor_kernels.hip
    python tools/isa_stats.py /tmp/q.s [substring of the mangled name] [--loops] [--dump-loop FILE]

--loops        list every loop (label, first line, last line, static instructions) instead of the widest only
--dump-loop F  write the widest loop's instructions of the (single) matching kernel to F
"""
import re
import sys

args = [x for x in sys.argv[1:] if not x.startswith("--")]
flags = [x for x in sys.argv[1:] if x.startswith("--")]
txt = open(args[0]).read()
want = args[1] if len(args) > 1 else ""
dump = args[2] if "--dump-loop" in flags and len(args) > 2 else None


def is_inst(l):
    return l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')


def classify(insts):
    c = dict(insts=len(insts), valu=0, f64=0, trans=0, readlane=0, writelane=0, salu=0, smem=0, lds=0, vmem=0, mfma=0, waitcnt=0, branch=0)
    for l in insts:
        op = l.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
            c['readlane'] += 1; c['valu'] += 1
        elif op.startswith('v_writelane'):
            c['writelane'] += 1; c['valu'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
            if '_f64' in op:
                c['f64'] += 1
            if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)_', op):
                c['trans'] += 1
        elif op.startswith('s_load') or op.startswith('s_buffer_load'):
            c['smem'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_cbranch') or op.startswith('s_branch'):
            c['branch'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
            c['vmem'] += 1
    return c


def loops_of(lines):
    """[(label, first index, last index)] of every backward branch target, widest first."""
    label_at = {}
    out = []
    for k, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            label_at[m.group(1)] = k
            continue
        m = re.match(r'\ts_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in label_at:
            out.append((m.group(1), label_at[m.group(1)], k))
    return sorted(out, key=lambda t: t[1] - t[2])


for f in re.split(r'\n(?=_ZN2qr11step_kernel\S*:)', txt):
    m = re.match(r'(_ZN2qr11step_kernel\S+):', f)
    if not m or want not in m.group(1):
        continue
    body = f.split('.Lfunc_end')[0]
    lines = body.splitlines()
    insts = [l for l in lines if is_inst(l)]
    tag = re.search(r'ILi(\d)E(\w)(\w)Li64EL[bi](\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E', m.group(1))
    name = "kind=%s %s%s TRAJ=%s ADAPT=%s POLICY=%s SINGLE=%s HELP=%s HREW=%s MAG=%s" % tag.groups() if tag else m.group(1)
    print(name, '| insts', len(insts), 'writelane', body.count('v_writelane'), 'readlane', body.count('v_readlane'),
          'scratch', body.count('scratch_'), 'vload', len(re.findall(r'(buffer|global)_load', body)),
          'vstore', len(re.findall(r'(buffer|global)_store', body)), 'waitcnt_vm', len(re.findall(r's_waitcnt.*vmcnt', body)))
    single = tag and tag.group(7) == '1'
    lp = loops_of(lines)
    if "--loops" in flags:
        for lab, a, b in lp:
            print('    loop', lab, 'lines', a, '-', b, classify([l for l in lines[a:b + 1] if is_inst(l)]))
    elif lp and not single:
        # the stepping wave's per-env-step loop: the widest loop that stores a done flag's worth of rows (the helper wave's loops
        # hold no buffer_store of the working set and are narrower)
        lab, a, b = lp[0]
        c = classify([l for l in lines[a:b + 1] if is_inst(l)])
        print('    step loop %s: static insts %d | valu %d (f64 %d, trans %d) readlane %d writelane %d | salu %d smem %d lds %d vmem %d mfma %d '
              'waitcnt %d branch %d' % (lab, c['insts'], c['valu'], c['f64'], c['trans'], c['readlane'], c['writelane'], c['salu'], c['smem'],
                                        c['lds'], c['vmem'], c['mfma'], c['waitcnt'], c['branch']))
        if dump:
            open(dump, 'w').write('\n'.join(lines[a:b + 1]) + '\n')
