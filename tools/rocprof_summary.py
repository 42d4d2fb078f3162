#!/usr/bin/env python
"""Turns a rocprofv3 `--kernel-trace --stats` result database (rocpd SQLite) into the CSV summary that is
committed under profiles/.   usage: python tools/rocprof_summary.py <results.db> <out.csv> [note...]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:110]


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = ' '.join(sys.argv[3:])
    con = sqlite3.connect(db)
    rows = con.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    with open(out, 'w', newline='') as fh:
        if note:
            fh.write(f'# {note}\n')
        fh.write('# source: rocprofv3 --kernel-trace --stats (durations in microseconds)\n')
        w = csv.writer(fh)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent'])
        for name, calls, total, avg, pct in rows:
            w.writerow([short(name), calls, round(total, 1), round(avg, 2), round(pct, 2)])
        conv = [(c, t, p) for n, c, t, a, p in rows if 'k_conv_igemm' in n]
        if conv:                                      # the five tile shapes of the one convolution kernel, together
            calls, total = sum(c for c, _, _ in conv), sum(t for _, t, _ in conv)
            w.writerow(['fiery::k_conv_igemm (all tile shapes)', calls, round(total, 1), round(total / calls, 2),
                        round(sum(p for _, _, p in conv), 2)])
    print(f'wrote {out} ({len(rows)} kernels)')


if __name__ == '__main__':
    main()
