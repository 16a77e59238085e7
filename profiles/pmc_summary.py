#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (one results db per pass)."""
import collections
import re
import sqlite3
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'^void\s+', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:60]


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        seen = set()
        for kname, cname, val, did, d in cur.execute(
                'select kernel_name, counter_name, value, dispatch_id, duration from counters_collection'):
            k = short(str(kname))
            a = acc[k][cname]
            a[0] += 1
            a[1] += float(val)
            if (path, did) not in seen:
                seen.add((path, did))
                dur[k][0] += 1
                dur[k][1] += float(d)
    counters = sorted({c for k in acc for c in acc[k]})
    for k in sorted(acc, key=lambda k: -dur[k][1]):
        print('%s  (avg %.1f us over %d dispatches)' % (k, dur[k][1] / dur[k][0] / 1e3, dur[k][0]))
        for c in counters:
            if c in acc[k]:
                n, s = acc[k][c]
                print('    %-28s %16.1f' % (c, s / n))


if __name__ == '__main__':
    main(sys.argv[1:])
