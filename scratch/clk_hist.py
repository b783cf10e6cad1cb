"""Per-chunk s_memtime stamps of k_threshold_lean (NFC_DEBUG_CLK=<file>): where do the waves spend their time, who is slow?"""
import sys
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4).astype(np.int64)
ok = h[:, 0] > 0
h = h[ok]
t0 = h[:, 0].min()
start, pro, loop, epi = h[:, 0] - t0, h[:, 1] - h[:, 0], h[:, 2] - h[:, 1], h[:, 3] - h[:, 2]
end = h[:, 3] - t0
tot = h[:, 3] - h[:, 0]
def q(a):
    return ' '.join('%8d' % v for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
print('%d chunks; percentiles 0 10 50 90 99 100 (ticks)' % len(h))
for name, a in (('start', start), ('incoming', pro), ('loop', loop), ('summary', epi), ('total', tot), ('end', end)):
    print('%-9s %s' % (name, q(a)))
c = np.nonzero(ok)[0]
blk = c // 4
for mod, label in ((8, 'block % 8 (XCD, if blocks go round-robin)'), (4, 'wave in block')):
    key = (blk % mod) if mod == 8 else (c % 4)
    print(label, ' '.join('%d:%.0f/%.0f' % (k, tot[key == k].mean(), end[key == k].max()) for k in range(mod)))
# slowest waves
idx = np.argsort(-end)[:12]
print('latest finishers: chunk, start, incoming, loop, summary, end')
for i in idx:
    print('  %5d %8d %8d %8d %8d %8d' % (c[i], start[i], pro[i], loop[i], epi[i], end[i]))
third = len(h) // 3
print('mean total by chunk-index third: %.0f %.0f %.0f' % (tot[:third].mean(), tot[third:2 * third].mean(), tot[2 * third:].mean()))
