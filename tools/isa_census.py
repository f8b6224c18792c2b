"""Per-phase instruction census of one kernel from the compiler's assembly.

    python tools/isa_census.py [--kernel interp_pool_f16x3_kernel] [--src ppsurf_amd/csrc/pps_decode.hip] [-D...] [--md out.md]

Compiles the source for gfx950 with line tables (-gline-tables-only: same code, plus `.loc` directives), takes the kernel's main loop (the
largest innermost region between a loop-header label and its back edge) and attributes every instruction to a PHASE by the source line of the
innermost inlined function it came from (PHASES below: line ranges of pps_decode.hip / pps_common.h, looked up by marker text so that edits do not
shift them), then counts by class: MFMA, other VALU (split further: conversions / packed / max3 / mov / arithmetic), LDS, vector memory, scalar,
waitcnt.  The counts are STATIC instructions of one trip of the loop = one tile of a wave (16 neighbour rows of one query)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only', '-S', '-gline-tables-only']


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def valu_kind(op):
    if op.startswith('v_cvt'):
        return 'cvt'
    if op.startswith('v_pk_'):
        return 'packed'
    if op.startswith(('v_max3', 'v_max_', 'v_min', 'v_maximum', 'v_minimum')):
        return 'max'
    if op.startswith(('v_mov', 'v_accvgpr', 'v_perm', 'v_swap', 'v_readlane', 'v_readfirstlane', 'v_writelane', 'v_permlane')):
        return 'move'
    if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')):
        return 'transcendental'
    if op.startswith(('v_cndmask', 'v_cmp')):
        return 'select'
    if '_dpp' in op or 'dpp' in op:
        return 'dpp'
    return 'arith'


def find_line(path, marker, nth=0):
    with open(path) as f:
        hits = [i + 1 for i, l in enumerate(f) if marker in l]
    if len(hits) <= nth:
        raise SystemExit('marker {!r} not found in {}'.format(marker, path))
    return hits[nth]


def phases_interp_f16x3(decode, common):
    """[(file, first line, last line, phase name)] for interp_pool_f16x3_kernel."""
    k0 = find_line(decode, 'void interp_pool_f16x3_kernel(')
    def d(marker, nth=0):
        with open(decode) as f:
            hits = [i + 1 for i, l in enumerate(f) if marker in l and i + 1 >= k0]
        return hits[nth]
    gather0 = d('const int64_t qi = ')
    split0 = d('x[kb] = split_f16_r(amax, a[2 * kb], a[2 * kb + 1]);')
    prio = d('__builtin_amdgcn_s_setprio(PPS_PRIO);')
    soft0 = d('// ---- softmax over the 64 neighbours')
    comb0 = d('float mw[4], sw[4];')
    attw0 = d('float an = 0.f;')
    join0 = d('// h3 back to fp32 (hi + lo)')
    pool0 = d('rows16_sum_transposed(a, lane);')
    end = d('range_commit(amax, range);')
    c = lambda m, nth=0: find_line(common, m, nth)
    return [
        (common, c('__device__ __forceinline__ HiLo split_f16('), c('// Range guard of the split') - 1, 'split (cvt_pkrtz hi, sub, cvt lo)'),
        (common, c('__device__ __forceinline__ void range_track('), c('__device__ __forceinline__ HiLo split_f16_r(') - 1, 'range guard (max3)'),
        (common, c('__device__ __forceinline__ void join_f16('), c('// out blocks OB0 .. OB0+NOB-1') - 1, 'join (hi + lo -> fp32)'),
        (common, c('f32x4 o0 = m0 + c0, o1 = m1 + c1;'), c('f32x4 o0 = m0 + c0, o1 = m1 + c1;'), 'accumulator join (main + correction)'),
        (common, c('for (int r = 0; r < 4; ++r) { o0[r] = fmaxf(o0[r], 0.f); o1[r] = fmaxf(o1[r], 0.f); }'),
         c('for (int r = 0; r < 4; ++r) { o0[r] = fmaxf(o0[r], 0.f); o1[r] = fmaxf(o1[r], 0.f); }'), 'ReLU of the layer output'),
        (common, c('template <int KB, int NOB, int ACT, bool FENCE = true, bool INIT = false, int NP = 3, class Sink, class Hook>'),
         c('sink(ob >> 1, o0, o1);'), 'dense layer loop (MFMA, fragment reads, bias)'),
        (common, c('__device__ __forceinline__ void xyz_blocks('), c('// ---- cooperative weight-chunk streaming') - 1, 'xyz layer (K = 3 MFMA)'),
        (common, c('// ---- cooperative weight-chunk streaming'), c('// XCD-aware persistent tile order') - 1, 'weight stream (global_load_lds, waits)'),
        (decode, gather0, split0 - 1, 'gather G rows + offset, ReLU'),
        (decode, split0, prio - 1, 'split of h1 (call site)'),
        (decode, prio, soft0 - 1, 'layer chain call sites (stream steps, sinks)'),
        (decode, soft0, comb0 - 1, 'softmax: row max / exp / row sum (DPP), stats to LDS'),
        (decode, comb0, attw0 - 1, 'softmax: combine the 4 waves, factor'),
        (decode, attw0, join0 - 1, 'attention weight of a row (mean over heads)'),
        (decode, join0, pool0 - 1, 'join call site + scale by the weight'),
        (decode, pool0, end - 1, 'pooling: transposed 16-row sums, LDS, store'),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', default='interp_pool_f16x3_kernel')
    ap.add_argument('--src', default=os.path.join(REPO, 'ppsurf_amd', 'csrc', 'pps_decode.hip'))
    ap.add_argument('--md', default=None)
    ap.add_argument('--keep', default=None, help='write the assembly here')
    args, extra = ap.parse_known_args()
    common = os.path.join(REPO, 'ppsurf_amd', 'csrc', 'pps_common.h')
    out = args.keep or os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + FLAGS + extra + ['-o', out, args.src], stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.normpath(os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2))
    start = next(i for i, l in enumerate(lines) if re.match(r'_Z\d+' + args.kernel + r'\w*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    # the main loop: from the loop-header label with the most instructions up to the LAST branch back to it
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    best = None
    for i, l in enumerate(body):
        m = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    lo, hi = best
    phases = phases_interp_f16x3(args.src, common) if 'interp_pool_f16x3' in args.kernel else []
    table = collections.OrderedDict((p[3], collections.Counter()) for p in phases)
    table['(other lines)'] = collections.Counter()
    vk = collections.OrderedDict((p[3], collections.Counter()) for p in phases)
    vk['(other lines)'] = collections.Counter()
    cur = (None, 0)
    # the .loc state at the loop head: the last .loc before it
    for i in range(lo, -1, -1):
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', body[i])
        if m:
            cur = (files.get(int(m.group(1))), int(m.group(2)))
            break
    for l in body[lo:hi + 1]:
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            cur = (files.get(int(m.group(1))), int(m.group(2)))
            continue
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if not m or l.strip().startswith(('.', ';')):
            continue
        op = m.group(1)
        name = '(other lines)'
        for f, a, b, nm in phases:
            if cur[0] and os.path.basename(cur[0]) == os.path.basename(f) and a <= cur[1] <= b:
                name = nm
                break
        c = classify(op)
        table[name][c] += 1
        if c == 'valu':
            vk[name][valu_kind(op)] += 1
    classes = ['mfma', 'valu', 'lds', 'vmem', 'salu', 'wait', 'other']
    kinds = ['cvt', 'packed', 'max', 'move', 'arith', 'select', 'transcendental', 'dpp']
    tot = collections.Counter()
    for c in table.values():
        tot.update(c)
    out_lines = ['| phase | ' + ' | '.join(classes) + ' | VALU by kind (' + ' / '.join(kinds) + ') |', '|---|' + '---|' * (len(classes) + 1)]
    for name, c in table.items():
        if sum(c.values()) == 0:
            continue
        out_lines.append('| {} | {} | {} |'.format(name, ' | '.join(str(c[k]) for k in classes), ' / '.join(str(vk[name][k]) for k in kinds)))
    allk = collections.Counter()
    for c in vk.values():
        allk.update(c)
    out_lines.append('| **total (one loop trip)** | {} | {} |'.format(' | '.join('**{}**'.format(tot[k]) for k in classes), ' / '.join(str(allk[k]) for k in kinds)))
    text = '\n'.join(out_lines)
    head = ('kernel `{}`: main loop = {} assembly lines ({} instructions), VALU : MFMA = {:.2f} (static, one trip = one 16-row tile of a wave)'
            .format(args.kernel, hi - lo + 1, sum(tot.values()), tot['valu'] / max(1, tot['mfma'])))
    print(head)
    print(text)
    if args.md:
        with open(args.md, 'w') as f:
            f.write(head + '\n\n' + text + '\n')


if __name__ == '__main__':
    main()
