"""Reads the per-chunk lines a -DNFC_EX_PRINTF build of the library prints (threshold_wg.hip.h: the re-runs that evaluate failed rounds
in place) from stdin and prints what they add up to: chunks re-run, rounds evaluated in place, trips of the mask iteration.
    NFC_AMD_LIB=<-DNFC_TEST_HOOKS -DNFC_EX_PRINTF build> python tools/stress_step.py stress_dropouts_steps 2>&1 | python tools/exrounds_summary.py"""
import re, sys, collections
pat = re.compile(r'chunk (\d+): (\d+) of (\d+) rounds exact \(codes (\w+), first (-?\d+) last (-?\d+), (\d+) trips\) good (\d) why (\d+)')
n = ex = rounds = trips = gave = 0
hist = collections.Counter()
other = []
for line in sys.stdin:
    m = pat.search(line)
    if not m:
        if ' ms ' in line:
            other.append(line.rstrip())
        continue
    n += 1
    ex += int(m.group(2)); rounds += int(m.group(3)); trips += int(m.group(7)); gave += 1 - int(m.group(8))
    hist[int(m.group(2))] += 1
print('\n'.join(other))
print('chunk evaluations by the in-place form: %d (gave up: %d); rounds: %d, of them evaluated in place: %d (%.1f %%); trips of the mask iteration: %d (%.2f per such round)'
      % (n, gave, rounds, ex, 100.0 * ex / max(rounds, 1), trips, trips / max(ex, 1)))
print('rounds evaluated in place per chunk evaluation: ' + ', '.join('%d: %d' % kv for kv in sorted(hist.items())))
