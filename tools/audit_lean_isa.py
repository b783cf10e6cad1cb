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

  * k_threshold_wg: scalar registers.  The kernel sits at the 106-SGPR ceiling and spills (.sgpr_spill_count is printed: spills on
    gfx950 are v_writelane / v_readlane of a spill VGPR -- vector issue slots, and the s_nop pads in front of whoever reads the reloaded
    register); what matters is WHERE: the head of a regular round -- from the take of a round's samples to the round's barrier, the
    part every round of every wave runs -- must hold no compiler-emitted v_readlane / v_writelane outside the block that flushes the
    plane staging ring (taken every eighth round of one wave in four, never with a chunk's planes staged whole).  The counts over the
    whole regular-round loop (cold forms included) are printed for the record.

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
    sp = re.search(r'\.sgpr_spill_count:\s+(\d+)', md)
    extra = ''
    if 'k_threshold_wg' in name:
        # the regular-round loop: its header is the labelled block that holds the first take (v_accvgpr_read of a0 / a8)
        take = next((i for i, l in enumerate(body) if re.search(r'v_accvgpr_read_b32 v\d+, a0\b', l)), None)
        if take is not None:
            h = take
            while h > 0 and not re.match(r'^\.LBB\d+_\d+:', body[h]):
                h -= 1
            hn = body[h].split(':')[0][2:]
            def spill_ops(lines):
                n = [0, 0, 0]
                in_a = False
                for l in lines:
                    if 'ASMSTART' in l:
                        in_a = True
                    elif 'ASMEND' in l:
                        in_a = False
                    elif not in_a:
                        c = l.split(';')[0]
                        n[0] += 'v_readlane_b32' in c
                        n[1] += 'v_writelane_b32' in c
                        n[2] += 's_nop' in c
                return n
            loop = [l for l in body if ('Header=' + hn + ' ') in l + ' ' or ('Parent Loop ' + hn + ' ') in l + ' ']
            # (every line of the loop: blocks are annotated, their instructions are not -- take the labelled blocks' extents)
            idx = [i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l) or re.match(r'^; %bb\.\d+:', l)]
            in_loop_lines = []
            for a, b2 in zip(idx, idx[1:] + [len(body)]):
                hd = body[a]
                if a == h or ('Header=' + hn + ' ') in hd + ' ' or ('Parent Loop ' + hn + ' ') in hd + ' ':
                    in_loop_lines += body[a:b2]
            whole = spill_ops(in_loop_lines)
            # the head: from the header to the round's barrier, minus the blocks that store to global memory (the staging ring's flush)
            bar = next(i for i in range(take, len(body)) if 's_barrier' in body[i])
            head = []
            for a, b2 in zip(idx, idx[1:] + [len(body)]):
                if a < h or a > bar:
                    continue
                blk = body[a:min(b2, bar)]
                if any('global_store' in l for l in blk):
                    continue
                # (the block in front of the flush loads its plane pointers: part of the flush when the next block stores)
                head += blk
            # a reload block that only feeds the flush: the one right before the storing block
            hd = spill_ops(head)
            flush_pre = 0
            for a, b2 in zip(idx, idx[1:] + [len(body)]):
                if h <= a <= bar and any('global_store' in l for l in body[a:b2]):
                    k = idx.index(a)
                    if k > 0:
                        flush_pre = spill_ops(body[idx[k - 1]:a])[0]
            hd[0] -= flush_pre
            extra = ', sgpr_spill_count %s, round head (to the barrier, flush aside): %d v_readlane %d v_writelane, whole round loop: %d / %d / %d s_nop' % (
                sp.group(1) if sp else '?', hd[0], hd[1], whole[0], whole[1], whole[2])
            # (four rows per step -- configs[1] / [2], every input kind: none at all.  Six and eight rows per step keep eight more ballots
            # alive and reload five or six loop invariants per 1 536 / 2 048-sample round -- the staging mode, the superstep's round
            # count, a lane mask: printed, 2 % of a round's vector instructions, not a finding)
            if (hd[0] or hd[1]) and 'ELi4ELi' in name:
                print('%s: spill traffic in the head of a regular round: %d v_readlane, %d v_writelane' % (name, hd[0], hd[1]))
                n_bad += 1
    print('%s: %d asm loads, %d counted waits, vgpr_count %s (agpr %s)%s, %d findings' % (name, n_loads, n_takes, vg.group(1) if vg else '?',
                                                                                        ag.group(1) if ag else '?', extra, n_bad))
    bad += n_bad
sys.exit(1 if bad else 0)
