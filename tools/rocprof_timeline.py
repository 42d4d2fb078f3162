#!/usr/bin/env python
"""Raw kernel timeline (start / end relative to the first row, microseconds) of a window of a rocprofv3 --kernel-trace database.
usage: python tools/rocprof_timeline.py <results.db> <substring> [first match index] [rows]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    needle = sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
    extra = ', queue_id' if 'queue_id' in cols else ', 0'
    extra += ', grid_size' if 'grid_size' in cols else (', grid_x' if 'grid_x' in cols else ', 0')
    rows = con.execute(f'select name, start, end{extra} from kernels order by start').fetchall()
    hits = [i for i, r in enumerate(rows) if needle in r[0]]
    if not hits:
        print('no match; columns:', cols)
        return
    i0 = hits[min(skip, len(hits) - 1)]
    t0 = rows[i0][1]
    for name, st, en, q, g in rows[i0:i0 + n]:
        short = name.split('(')[0][-60:]
        print(f'{(st - t0) / 1e3:9.1f} -> {(en - t0) / 1e3:9.1f}  ({(en - st) / 1e3:7.1f} us)  queue {q}  grid {g}  {short}')


if __name__ == '__main__':
    main()
