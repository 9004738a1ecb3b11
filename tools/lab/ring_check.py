"""Static check of the residual-ring registers of gemm_blk16_kernel<.., EPI 2> in hipcc's ISA output (tools/lab: build-time aid, not product code).

The ring's loads are inline asm, invisible to hipcc's waitcnt pass: a register COPY (v_mov / v_accvgpr) of a ring register between its load and the
hand-counted wait would copy data that is still in flight.  This lists every instruction that READS a ring register other than the epilogue's
adds, per kernel, with the line numbers of the ring loads for orientation.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only w-hmr_amd/csrc/gemm_blk.hip -o /tmp/gemm_blk.s
    python tools/lab/ring_check.py /tmp/gemm_blk.s
"""
import re
import sys


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def main(path):
    txt = open(path).read().splitlines()
    starts = [i for i, l in enumerate(txt) if re.match(r'^_Z17gemm_blk16_kernelILi\dELi\dELi2ELi1EEv', l)]      # the shipped schedule
    bad_total = 0
    for s in starts:
        e = next(i for i in range(s, len(txt)) if 's_endpgm' in txt[i])
        body = txt[s:e]
        ring, in_asm = set(), False
        loads = []
        for n, l in enumerate(body):
            if '#ASMSTART' in l:
                in_asm = True
            elif '#ASMEND' in l:
                in_asm = False
            elif in_asm and 'global_load_dwordx4' in l:
                ring |= regs(l.split()[1].rstrip(','))
                loads.append(n)
        bad = []
        for n, l in enumerate(body):
            t = l.split()
            if not t or not t[0].startswith('v_mov'):
                continue
            ops = [x.rstrip(',') for x in t[1:]]
            if len(ops) >= 2 and regs(ops[1]) & ring and loads and n > loads[0]:
                bad.append((n, l.strip()))           # behind the first ring load in layout order (before it the registers hold other values / zeros)
        name = txt[s].split(':')[0]
        print('%s: ring registers %d, ring loads at %d sites (first %s, last %s), v_mov READING a ring register: %d'
              % (name, len(ring), len(loads), loads[:1], loads[-1:], len(bad)))
        for n, l in bad[:40]:
            print('    line +%d: %s' % (n, l))
        bad_total += len(bad)
    return bad_total


if __name__ == '__main__':
    sys.exit(1 if main(sys.argv[1]) else 0)
