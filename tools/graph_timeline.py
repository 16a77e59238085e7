"""Timeline of the kernels of HIP-graph replays from a rocprofv3 --kernel-trace database: per kernel the average duration and
the average gap to the previous kernel's end, and the busy / idle split of a step.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d out -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0
                 python3 tools/graph_timeline.py out/kt_results.db"""
import collections, re, sqlite3, sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t] or [t for t in tabs if 'kernel_dispatch' in t]
sym = [t for t in tabs if 'kernel_symbol' in t]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd[0])]
scols = [r[1] for r in cur.execute('pragma table_info(%s)' % sym[0])]
namecol = 'kernel_name' if 'kernel_name' in scols else ('display_name' if 'display_name' in scols else scols[-1])
names = dict(cur.execute('select id, %s from %s' % (namecol, sym[0])))
rows = list(cur.execute('select kernel_id, start, end from %s order by start' % kd[0]))


def short(n):
    n = str(n).replace('(anonymous namespace)::', '')
    n = re.sub(r'^void\s+', '', n)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', n)
    return (m.group(1) if m else n)[:60]


last = int(sys.argv[2]) if len(sys.argv) > 2 else 37 * 20          # the timed replays are at the end of the trace
rows = rows[-last:]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
busy = idle = 0
for i, (k, s, e) in enumerate(rows):
    n = short(names.get(k, k))
    dur[n].append(e - s)
    busy += e - s
    if i:
        g = s - rows[i - 1][2]
        if g < 200000:            # not the gap between replays / host stalls
            gap[n].append(g)
            idle += max(g, 0)
print('%-62s %6s %9s %9s' % ('kernel', 'calls', 'avg us', 'gap us'))
for n in sorted(dur, key=lambda n: -sum(dur[n])):
    print('%-62s %6d %9.2f %9.2f' % (n, len(dur[n]), sum(dur[n]) / len(dur[n]) / 1e3, (sum(gap[n]) / len(gap[n]) / 1e3) if gap[n] else 0))
print('kernels busy %.1f us, gaps %.1f us over %d dispatches (span %.1f us)' % (busy / 1e3, idle / 1e3, len(rows), (rows[-1][2] - rows[0][1]) / 1e3))
