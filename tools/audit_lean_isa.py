#!/usr/bin/env python3
"""Audit of k_threshold_lean's ISA (cdna_hip_programming.md 5.7 item 4): the registers the asm loads fill must be touched by
nothing except those loads, the hand-counted waits and the envelope arithmetic that follows a wait -- a compiler copy or spill
of such a register between a load and its wait would read a register that is still being loaded.

usage: tools/audit_lean_isa.py <file.s>      (hipcc -S --cuda-device-only output)"""
import re
import sys

text = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r'^(_ZN3nfc16k_threshold_lean\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', text, re.S | re.M):
    name, body = m.group(1), m.group(2).split('\n')
    # the asm statements' loads
    loads = [(i, l) for i, l in enumerate(body) if 'global_load' in l and i > 0 and 'ASMSTART' in body[i - 1]]
    if not loads:
        print(name, 'no asm loads'); bad += 1; continue
    regs = set()
    for _, l in loads:
        d = re.search(r'global_load_\w+ (v\[(\d+):(\d+)\]|v(\d+)),', l)
        if d.group(2):
            regs.update(range(int(d.group(2)), int(d.group(3)) + 1))
        else:
            regs.add(int(d.group(4)))
    last = loads[-1][0]
    # the audited stretch: the loop, from its first hand-written wait to the drain behind it (the first hand-written vmcnt(0)
    # after the last asm load)
    waits = [i for i, l in enumerate(body) if 's_waitcnt vmcnt(' in l and 'ASMSTART' in body[i - 1]]
    pf = len(loads) // 8          # 4 loads per step, once in the prologue and once in the loop
    first = waits[pf]             # the loop's first wait (the prologue's waits, then the first allowance reading the samples, come before)
    drains = [i for i in waits if i > last and 'vmcnt(0)' in body[i]]
    end = drains[0] if drains else len(body)
    def touched(l):
        out = set()
        for a, b in re.findall(r'v\[(\d+):(\d+)\]', l):
            out.update(range(int(a), int(b) + 1))
        for a in re.findall(r'\bv(\d+)\b', l):
            out.add(int(a))
        return out
    n_bad = 0
    released = False   # between a hand-written wait and the next asm load the step's registers may be READ
    for i in range(first, end):
        l = body[i]
        if not l.startswith('\t') or l.strip().startswith(';'):
            continue
        in_asm = 'ASMSTART' in body[i - 1]
        if in_asm and 's_waitcnt vmcnt(' in l:
            released = True
            continue
        if in_asm and 'global_load' in l:
            released = False
            continue
        t = touched(l) & regs
        if not t:
            continue
        dst = touched(l.split(',')[0].split(None, 1)[1]) if ' ' in l.strip() else set()
        if released and not (dst & regs):
            continue   # a read after the wait, before the registers go back to the loads
        print('%s: line %d touches a load register%s: %s' % (name, i, '' if released else ' while its load may be in flight', l.strip()))
        n_bad += 1
    print('%s: %d asm loads into %d registers, loop lines %d..%d, %d suspicious' % (name, len(loads), len(regs), first, end, n_bad))
    bad += n_bad
sys.exit(1 if bad else 0)
