#!/usr/bin/env python
"""Prints per-kernel averages of the PMC counters in a rocprofv3 result database (rocpd SQLite)."""
import collections
import glob
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:70]


def main():
    for db in sorted(glob.glob(sys.argv[1], recursive=True)):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        views = [t for t in tabs if 'counter' in t.lower() or 'pmc' in t.lower()]
        print('==', db, views)
        if 'counters_collection' in tabs:
            cols = [r[1] for r in con.execute('pragma table_info(counters_collection)')]
            kname = 'kernel_name' if 'kernel_name' in cols else 'name'
            cname = 'counter_name' if 'counter_name' in cols else 'counter'
            agg = collections.defaultdict(lambda: [0, 0.0])
            for name, counter, value in con.execute(f'select {kname}, {cname}, value from counters_collection'):
                a = agg[(short(name), counter)]
                a[0] += 1
                a[1] += value
            for (name, counter), (n, total) in sorted(agg.items()):
                print(f'{name:72s} {counter:20s} n={n:4d} avg={total / n:16.1f} total={total:18.1f}')
        else:
            for t in views[:6]:
                print(t, [r[1] for r in con.execute(f'pragma table_info({t})')])
                for row in con.execute(f'select * from {t} limit 3'):
                    print('   ', row)


if __name__ == '__main__':
    main()
