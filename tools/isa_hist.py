"""tools/isa_hist.py — static instruction mix of the gfx950 kernels of libqmhip from the compiler's assembly (no GPU needed): per kernel the counts by class
(FP64 VALU, other VALU, MFMA, SALU, LDS, vector memory) and, with --lines, the source lines that own the most instructions (hipcc -gline-tables-only).
The dynamic mix (SQ PMC counters, profiles/flops_pmc.json) says how often; this says WHERE.
K1b is compiled with QM_LQ_RB_ONLY=1: without the dense R0 (u - u_nom) path a wave does not execute with the shipped block-diagonal input weight, i.e. the executed stream.
usage: python tools/isa_hist.py [--other] [--lines N] [kernel ...]      (default kernels: qm_lq_kernel qm_riccati_kernel qm_wbc_kernel)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qm_control_amd.build_flags import HIPCC_FLAGS, DEVICE_UNITS      # the product build's flags


def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_') and 'f64' in op: return 'valu_f64'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'


def main(argv):
    nlines = 0; by_other = False
    if argv and argv[0] == '--other': by_other = True; argv = argv[1:]          # rank the lines by their non-FP64 VALU instructions instead of by all instructions
    if argv and argv[0] == '--lines': nlines = int(argv[1]); argv = argv[2:]
    kernels = argv or ['qm_lq_kernel', 'qm_riccati_kernel', 'qm_wbc_kernel']
    with tempfile.TemporaryDirectory() as d:
        units = []
        for src, extra in DEVICE_UNITS:      # every translation unit with its own flags (qm_control_amd/build_flags.py)
            stem = os.path.splitext(os.path.basename(src))[0]
            subprocess.check_call(['/opt/rocm/bin/hipcc'] + HIPCC_FLAGS + extra + (['-gline-tables-only'] if nlines else []) +
                                  ['-DQM_LQ_RB_ONLY=1', '-I' + os.path.join(ROOT, 'include'), '--save-temps', '-c', os.path.join(ROOT, src), '-o', os.path.join(d, stem + '.o')], cwd=d, stderr=subprocess.DEVNULL)
            units.append(open(os.path.join(d, stem + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read())
    for name in kernels:
        m = None
        for s in units:
            m = re.search(r'^(_Z\d+%s\w*):[^\n]*\n(.*?)\n\.Lfunc_end' % name, s, re.S | re.M)
            if m: break
        if not m: print(name, 'not found'); continue
        files = {int(f.group(1)): (f.group(3) or f.group(2)).split('/')[-1] for f in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
        cls = collections.Counter(); ops = collections.Counter(); per = collections.Counter(); perop = collections.defaultdict(collections.Counter); cur = None
        for line in m.group(2).split('\n'):
            t = line.strip()
            if t.startswith('.loc'): p = t.split(); cur = (files.get(int(p[1]), p[1]), int(p[2])); continue
            if not t or t.startswith(('.', ';', '//')) or t.endswith(':'): continue
            op = t.split()[0]; cls[classify(op)] += 1; ops[op] += 1; per[cur] += 1; perop[cur][op] += 1
        tot = sum(cls.values()); valu = cls['valu_f64'] + cls['valu_other'] + cls['mfma']
        print('%s: %d instructions %s; non-FP64 share of VALU %.2f' % (name, tot, dict(cls), cls['valu_other'] / max(1, valu)))
        print('   top opcodes: ' + ' '.join('%s:%d' % x for x in ops.most_common(14)))
        if by_other:
            per = collections.Counter({k: sum(n for op, n in perop[k].items() if classify(op) == 'valu_other') for k in perop})
        for k, n in per.most_common(nlines):
            print('   %5d %s:%d  %s' % (n, k[0] if k else None, k[1] if k else 0, ' '.join('%s:%d' % x for x in perop[k].most_common(5))))


if __name__ == '__main__':
    main(sys.argv[1:])
