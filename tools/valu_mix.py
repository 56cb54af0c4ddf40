"""Average issue cost of a kernel's vector instructions, from its own instruction mix and the measured per-instruction rates
(profiles/r06_valu_rate.txt, made by tools/ubench/valu_rate.hip on the GPU box).

    python tools/valu_mix.py                      -> profiles/r06_valu_mix.txt (every kernel a valu_roofline of bench.py names)
    python tools/valu_mix.py FILE.hip SUBSTRING   -> one kernel

What the rate table says about gfx950: a SIMD issues a wave64 vector instruction in 2 cycles only for the plain 32-bit VOP2 forms
(v_add_u32, v_sub_u32, v_and/or/xor_b32, and the 16-bit v_max/min_i16 ...), in about 3.5 for v_fma_f32 / v_bitop3_b32, and in 4 for
everything the DP kernels are made of: every packed-16 op (v_pk_*), every DPP-modified op, v_max/min_i32, the three-operand integer ops
(v_max3, v_med3, v_add3, v_lshl_add, v_and_or, v_bfi, v_alignbit, v_perm), shifts, compares, carries, multiplies.  MI355X_MICROARCH.md's
"2 cycles" is the first class; its own `v_fma_f32` row reads 3.5 here with eight waves per SIMD, 4 for one wave alone.

The mix is STATIC: the instructions of the kernel's largest loop nest (from the first backward-branch target to the last backward branch),
each counted once.  That is the row step / column step for every kernel measured here, whose time is that loop."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ciri_long_amd', 'csrc')

# cycles per wave64 instruction per SIMD at 8 resident waves (profiles/r06_valu_rate.txt), by mnemonic prefix; default 4
FULL = ('v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_max_i16', 'v_min_i16', 'v_max_u16', 'v_min_u16',
        'v_add_u16', 'v_sub_u16', 'v_mov_b32', 'v_not_b32', 'v_add_f32', 'v_mul_f32')
MID = ('v_bitop3_b32', 'v_fma_f32', 'v_fmac_f32')


def cost(mn, text):
    if 'dpp' in text or 'sdwa' in text:
        return 4.0
    if mn.endswith('_e64'):
        mn = mn[:-4]
    if mn.endswith('_e32'):
        mn = mn[:-4]
    if mn in MID:
        return 3.5
    if mn in FULL and ' clamp' not in text.replace('clamp', ' clamp'):
        return 2.0
    if mn in ('v_sub_u32', 'v_add_u32'):      # clamp forms measured at 2.2
        return 2.2
    return 4.0


def kernel_asm(hip_file, needle, defines=()):
    with tempfile.TemporaryDirectory(dir='/tmp') as d:
        out = os.path.join(d, 'k.s')
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-I', CSRC, '-I', os.path.join(ROOT, 'include')] + list(defines) + [hip_file, '-o', out]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        text = open(out).read()
    funcs = re.split(r'\n(?=[A-Za-z_][\w.$]*:\s*(?:;.*)?\n)', text)
    best = None
    for f in funcs:
        name = f.split(':', 1)[0]
        if needle in name and 's_endpgm' in f and (best is None or len(f) > len(best[1])):
            best = (name, f)
    if best is None:
        raise SystemExit('no kernel matching %r in %s' % (needle, hip_file))
    return best


def loop_body(body):
    lines = body.split('\n')
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r'^(\.LBB[\w]+):', ln)] if m}
    lo, hi = None, None
    for i, ln in enumerate(lines):
        m = re.search(r'\bs_cbranch_\w+\s+(\.LBB\w+)|\bs_branch\s+(\.LBB\w+)', ln)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:              # backward branch
                lo = labels[tgt] if lo is None else min(lo, labels[tgt])
                hi = i if hi is None else max(hi, i)
    return lines if lo is None else lines[lo:hi + 1]


def mix(hip_file, needle, defines=()):
    name, body = kernel_asm(hip_file, needle, defines)
    counts, cyc, n = {}, 0.0, 0
    salu = mem = 0
    for ln in loop_body(body):
        ln = ln.strip()
        m = re.match(r'^(v_\w+)\b(.*)', ln)
        if m:
            mn = m.group(1)
            c = cost(mn, m.group(2))
            counts[(mn, c)] = counts.get((mn, c), 0) + 1
            cyc += c; n += 1
        elif re.match(r'^s_\w+', ln) and not ln.startswith('s_waitcnt') and not ln.startswith('s_nop'):
            salu += 1
        elif re.match(r'^(ds_|global_|buffer_|flat_|scratch_)', ln):
            mem += 1
    return name, n, cyc / max(n, 1), counts, salu, mem


KERNELS = [('K3 poa_consensus_kernel', 'ccs_poa.hip', 'poa_consensus_kernel', ()),
           ('K1w ssw_scanw_kernel (gap_open == gap_extend)', 'ssw_scan_wide.hip', 'ssw_scanw_kernelILb1', ()),
           ('K1w ssw_scanw_kernel (gap_open > gap_extend)', 'ssw_scan_wide.hip', 'ssw_scanw_kernelILb0', ()),
           ('K1s ssw_scan_kernel', 'ssw_scan.hip', 'ssw_scan_kernel', ()),
           ('prefilter ssw_prefilter_kernel', 'ssw_prefilter.hip', 'ssw_prefilter_kernel', ()),
           ('K1l ssw_lanes_kernel', 'ssw_lanes.hip', 'ssw_lanes_kernel', ())]


def main():
    if len(sys.argv) >= 3:
        todo = [(sys.argv[2], sys.argv[1], sys.argv[2], ())]
        out = sys.stdout
    else:
        todo = [k for k in KERNELS if os.path.exists(os.path.join(CSRC, k[1]))]
        out = open(os.path.join(ROOT, 'profiles', 'r06_valu_mix.txt'), 'w')
        out.write('# tools/valu_mix.py: vector instructions of each kernel\'s largest loop nest, weighted by the measured issue cost per instruction\n'
                  '# (profiles/r06_valu_rate.txt: 2 cycles plain 32-bit VOP2, 3.5 v_bitop3 / v_fma, 4 everything else) -> average cycles per vector instruction\n')
    for title, f, needle, defs in todo:
        path = f if os.path.isabs(f) or os.path.exists(f) else os.path.join(CSRC, f)
        try:
            name, n, avg, counts, salu, mem = mix(path, needle, defs)
        except SystemExit as e:
            out.write('%s: %s\n' % (title, e))
            continue
        by = {}
        for (mn, c), k in counts.items():
            by[c] = by.get(c, 0) + k
        out.write('%-52s %5d vector instr in the loop nest (+ %d scalar, %d memory): avg %.2f cycles  [%s]  %s\n'
                  % (title, n, salu, mem, avg, ', '.join('%d at %.1f' % (k, c) for c, k in sorted(by.items())), name[:60]))
        top = sorted(counts.items(), key=lambda x: -x[1])[:12]
        out.write('    most frequent: %s\n' % ', '.join('%s x%d' % (mn, k) for (mn, c), k in top))
    if out is not sys.stdout:
        out.close()
        print(open(os.path.join(ROOT, 'profiles', 'r06_valu_mix.txt')).read())


if __name__ == '__main__':
    main()
