#!/usr/bin/env python3
"""Print one step's kernel timeline (start offset, duration, gap before) from a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps begin at k_fill
starts = [i for i, r in enumerate(rows) if 'k_fill' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[k], starts[k + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += e - s
    print('%8.1f us  dur %7.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'].replace('nfc::', '')[:70]))
    prev_end = e
period = (int(rows[b]['Start_Timestamp']) - t0) / 1e3
print('step period %.1f us, busy %.1f us, last kernel end at %.1f us' % (period, busy / 1e3, (prev_end - t0) / 1e3))
