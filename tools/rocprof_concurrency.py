#!/usr/bin/env python
"""How busy a rocprofv3 --kernel-trace window is: union of the kernel intervals / wall time, and the share of the wall time with
0 / 1 / 2 / 3+ kernels in flight; the longest idle gaps with the kernels around them.
usage: python tools/rocprof_concurrency.py <results.db> [skip_first_fraction=0.3]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = con.execute('select name, start, end from kernels order by start').fetchall()
    rows = [r for r in rows if 'fiery::' in r[0] and 'pack' not in r[0]]
    t0, t1 = rows[0][1], rows[-1][2]
    lo = t0 + skip * (t1 - t0)
    rows = [r for r in rows if r[1] >= lo]
    ev = []
    for n, s, e in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, ev[0][0], {}
    for t, d in ev:
        hist[min(level, 3)] = hist.get(min(level, 3), 0) + (t - last)
        level += d
        last = t
    wall = ev[-1][0] - ev[0][0]
    print(f'{len(rows)} kernels over {wall / 1e6:.2f} ms: in flight 0 / 1 / 2 / 3+ kernels for '
          + ' / '.join(f'{hist.get(k, 0) / wall:.1%}' for k in range(4)) + f'; kernel-time sum / wall = {sum(e - s for _, s, e in rows) / wall:.2f}')
    gaps = []
    cur_end, cur_name = rows[0][2], rows[0][0]
    for n, s, e in rows[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, cur_name, n))
        if e > cur_end:
            cur_end, cur_name = e, n
    gaps.sort(reverse=True)
    print(f'idle gaps: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e3:.1f} us; the longest:')
    for g, a, b in gaps[:8]:
        print(f'  {g / 1e3:7.1f} us  after {a.split("(")[0][-50:]}  before {b.split("(")[0][-50:]}')


if __name__ == '__main__':
    main()
