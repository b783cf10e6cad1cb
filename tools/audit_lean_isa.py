#!/usr/bin/env python3
"""Audit of the ISA of k_threshold_lean and k_threshold_wg (cdna_hip_programming.md 5.7 item 4).  The kernels keep the samples
asked for ahead in accumulator registers they name literally (threshold_lean.hip.h, threshold_wg.hip.h); that is only sound
while the compiler itself never touches the accumulator file, i.e. while it neither spills vector registers nor uses scratch:

  * no v_accvgpr_* and no access to a[..] outside the kernel's own asm statements (;;#ASMSTART .. ;;#ASMEND),
  * .vgpr_spill_count 0 and no scratch / buffer access,
  * every hand-written wait in the loop is followed, inside the same statement, by the reads it releases,
  * k_threshold_wg: a statement whose loads take their base from a scalar register pair starts with wait states (s_nop 4): the
    pair may have been written by a vector instruction right in front of it (v_readlane of a spilled register), and a vector
    memory instruction that reads it within five issue slots gets the old value -- the compiler pads only what it emitted.

usage: tools/audit_lean_isa.py <file.s>      (hipcc -S --cuda-device-only output); exit status 1 on any finding"""
import re
import sys

text = open(sys.argv[1]).read()
bad = 0
meta = {}
for m in re.finditer(r'\.name:\s+(_ZN3nfc1[46]k_threshold_(?:lean|wg)\w+)\n(.*?)\.wavefront_size', text, re.S):
    meta[m.group(1)] = m.group(2)
for m in re.finditer(r'^(_ZN3nfc1[46]k_threshold_(?:lean|wg)\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', text, re.S | re.M):
    name, body = m.group(1), m.group(2).split('\n')
    in_asm = False
    n_bad = n_loads = n_takes = 0
    for i, l in enumerate(body):
        if 'ASMSTART' in l:
            in_asm = True
            first_in_asm = True
            continue
        if 'ASMEND' in l:
            in_asm = False
            continue
        code = l.split(';')[0]
        if in_asm:
            if code.strip():
                if first_in_asm and 'global_load' in code and re.search(r',\s*s\[\d+:\d+\]', code):
                    print('%s: line %d: scalar-base load at the head of an asm statement without wait states: %s' % (name, i, l.strip()))
                    n_bad += 1
                first_in_asm = False
            n_loads += 'global_load' in code
            n_takes += 's_waitcnt vmcnt' in code
            continue
        if re.search(r'v_accvgpr|\ba\[?\d+', code) or 'scratch_' in code or re.search(r'buffer_(load|store)', code):
            print('%s: line %d: the compiler touches the accumulator file / scratch: %s' % (name, i, l.strip()))
            n_bad += 1
    md = meta.get(name, '')
    for key in ('.vgpr_spill_count',):   # (a private segment may be reserved for scalar spill slots without ever being accessed: instructions are what counts)
        v = re.search(re.escape(key) + r':\s+(\d+)', md)
        if v is None or int(v.group(1)) != 0:
            print('%s: %s is %s' % (name, key, v.group(1) if v else 'missing'))
            n_bad += 1
    vg = re.search(r'\.vgpr_count:\s+(\d+)', md)
    ag = re.search(r'\.agpr_count:\s+(\d+)', md)
    print('%s: %d asm loads, %d counted waits, vgpr_count %s (agpr %s), %d findings' % (name, n_loads, n_takes, vg.group(1) if vg else '?',
                                                                                      ag.group(1) if ag else '?', n_bad))
    bad += n_bad
sys.exit(1 if bad else 0)
