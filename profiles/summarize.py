#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats results database into a small text summary
(per-kernel calls / total / average duration in microseconds / share)."""
import re
import sqlite3
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'^void\s+', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:100]


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    print('%-70s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'share%'))
    for name, calls, total, avg, pct in rows:
        print('%-70s %8d %12.1f %10.2f %7.2f' % (short(str(name)), calls, total / 1e3 if total > 1e6 else total, avg / 1e3 if total > 1e6 else avg, pct))


if __name__ == '__main__':
    main(sys.argv[1])
