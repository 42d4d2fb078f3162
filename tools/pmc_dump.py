#!/usr/bin/env python
"""Prints per-kernel averages of the PMC counters in a rocprofv3 result database (rocpd SQLite)."""
import collections
import glob
import sqlite3
import sys


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
                a = agg[(name.split('(')[0][-60:], counter)]
                a[0] += 1
                a[1] += value
            for (name, counter), (n, total) in sorted(agg.items()):
                print(f'{name:62s} {counter:28s} n={n:4d} avg={total / n:16.1f}')
        else:
            for t in views[:6]:
                print(t, [r[1] for r in con.execute(f'pragma table_info({t})')])
                for row in con.execute(f'select * from {t} limit 3'):
                    print('   ', row)


if __name__ == '__main__':
    main()
