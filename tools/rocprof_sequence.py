#!/usr/bin/env python
"""Prints which kernels run right before / after every dispatch of a given kernel in a rocprofv3 --kernel-trace database.
usage: python tools/rocprof_sequence.py <results.db> <substring of the kernel name>"""
import collections
import sqlite3
import sys


def main():
    db, needle = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    tables = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    view = 'kernels' if 'kernels' in tables else None
    if view is None:
        print('tables:', tables)
        return
    cols = [r[1] for r in con.execute(f'pragma table_info({view})')]
    rows = con.execute(f'select name, start, end, queue_id, stream_id from {view} order by start').fetchall() \
        if 'stream_id' in cols else con.execute(f'select name, start, end, queue_id, 0 from {view} order by start').fetchall()
    before, after = collections.Counter(), collections.Counter()
    gaps = []
    for i, (name, st, en, q, s) in enumerate(rows):
        if needle in name:
            if i > 0:
                before[rows[i - 1][0][:70]] += 1
            if i + 1 < len(rows):
                after[rows[i + 1][0][:70]] += 1
                gaps.append((rows[i + 1][1] - en) / 1e3)
    print(f'{sum(before.values())} dispatches of *{needle}*; kernel before:')
    for k, v in before.most_common(8):
        print(f'  {v:5d}  {k}')
    print('kernel after:')
    for k, v in after.most_common(8):
        print(f'  {v:5d}  {k}')
    if gaps:
        gaps.sort()
        print(f'gap to the next kernel: median {gaps[len(gaps) // 2]:.2f} us')


if __name__ == '__main__':
    main()
