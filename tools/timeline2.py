#!/usr/bin/env python3
"""Kernel timeline of a few steps from a rocprofv3 kernel trace csv, one column per HIP stream / queue (batches submitted
ahead run their threshold stage on a second stream): start offset, duration, gap to the previous kernel of the same queue.
usage: tools/timeline2.py <kernel_trace.csv> [first_fill_index] [n_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
fills = [i for i, r in enumerate(rows) if 'k_fill' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(fills) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 30
a = fills[k]
t0 = int(rows[a]['Start_Timestamp'])
last_end = {}
for r in rows[a:a + cnt]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r[qkey] if qkey else '0'
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print('%8.1f us  dur %7.1f  gap %6.1f  q %-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r['Kernel_Name'].replace('nfc::', '').replace('void ', '')[:60]))
