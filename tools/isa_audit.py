#!/usr/bin/env python3
"""Audit of the hand-pinned weight ring in the generated gfx950 assembly.

The weight prefetch ring of csrc/mlp_core.h issues its loads from `asm volatile`, which hipcc
treats as opaque: the destination VGPRs count as written at ;;#ASMEND, so the compiler is free
to copy, spill or re-use them BEFORE the data has landed (cdna_hip_programming.md 5.7 item 1).
This tool replays each kernel's instruction stream, models the in-order VMEM queue
(every load/store pushes one entry, `s_waitcnt vmcnt(N)` retires all but the N youngest) and
reports any instruction that references a VGPR which an asm load still has in flight.
It also reports the shape of the stream: MFMAs, ring loads, counted waits, full drains.

usage: isa_audit.py kernel.s   (from: hipcc -S --cuda-device-only ...)"""
import re
import sys

VREG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
VMEM = re.compile(r'^(global_|buffer_|scratch_|flat_)(load|store|atomic)')


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(asm):
    for m in re.finditer(r'^(\w+):\s*; @\1\s*$', asm, re.M):
        end = asm.find('s_endpgm', m.end())
        yield m.group(1), asm[m.end():end]


def audit(body):
    pending = []            # [(is_asm, dest_regs, line_no)] oldest first
    in_asm = False
    stats = dict(mfma=0, ring_loads=0, counted_waits=0, drains=0, other_vmem=0, scratch=0)
    bad = []
    for ln, raw in enumerate(body.split('\n')):
        line = raw.split(';')[0].strip() if not raw.strip().startswith(';;#') else raw.strip()
        if raw.strip().startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if raw.strip().startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not line or line.endswith(':') or line.startswith('.'):
            continue
        op = line.split()[0]
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', line)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n and n < len(pending) else ([] if n == 0 else pending)
                stats['counted_waits' if n else 'drains'] += 1
            continue
        used = vregs(line)
        if not in_asm:      # the asm load itself may legally re-target a slot whose previous data was waited for
            for is_asm, regs, at in pending:
                if is_asm and used & regs:
                    bad.append((ln, line, sorted(used & regs), at))
        if VMEM.match(op):
            dest = set()
            if 'load' in op and '_lds_' not in op and not line.rstrip().endswith(' lds'):
                dest = vregs(line.split(',')[0])       # (LDS-DMA has no register destination)
            pending.append((in_asm, dest if in_asm else set(), ln))
            if in_asm:
                stats['ring_loads'] += 1
            else:
                stats['other_vmem'] += 1
                stats['scratch'] += op.startswith('scratch_')
        elif op.startswith('v_mfma'):
            stats['mfma'] += 1
    return stats, bad


def main(path):
    asm = open(path).read()
    rc = 0
    for name, body in kernels(asm):
        stats, bad = audit(body)
        print(f"{name}: {stats}  violations={len(bad)}")
        for ln, line, regs, at in bad[:12]:
            print(f"    line {ln}: `{line}` touches v{regs} still in flight from the asm load at line {at}")
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
