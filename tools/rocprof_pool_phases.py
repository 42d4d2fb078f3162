#!/usr/bin/env python
"""Per pooling op in a rocprofv3 --kernel-trace database: prepass duration, gap, pooling-kernel duration, op span; grouped by the
kernel that ran right before the prepass (another pooling op = warm / back to back, a copy = cold).
usage: python tools/rocprof_pool_phases.py <results.db>"""
import collections
import sqlite3
import sys

import numpy as np


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute('select name, start, end from kernels order by start').fetchall()
    groups = collections.defaultdict(list)
    i = 0
    while i < len(rows):
        name, st, en = rows[i]
        if 'k_rank_columns' in name:
            j = i + 1
            pool = []
            while j < len(rows) and ('k_voxel_pool' in rows[j][0] or 'k_rank_columns' in rows[j][0]) and rows[j][1] < en + 400_000:
                if 'k_voxel_pool' in rows[j][0]:
                    pool.append(rows[j])
                if len(pool) and 'k_rank_columns' in rows[j][0] and rows[j][1] > pool[-1][2]:
                    break
                j += 1
            if pool and len(pool) == 1:
                before = rows[i - 1][0] if i > 0 else 'start'
                key = 'after a pooling kernel' if 'k_voxel_pool' in before else ('after a copy' if 'opy' in before or 'elementwise' in before else 'after ' + before[:40])
                idle = (st - rows[i - 1][2]) / 1e3 if i > 0 else 0.0
                groups[key].append(((en - st) / 1e3, (pool[0][1] - en) / 1e3, (pool[0][2] - pool[0][1]) / 1e3, (pool[0][2] - st) / 1e3, idle))
                i = j
                continue
        i += 1
    print(f'{"preceded by":32s} {"n":>4s}  prepass  gap   pooling   op span   idle before   (medians, us)')
    for k, v in groups.items():
        a = np.array(v)
        m = np.median(a, axis=0)
        print(f'{k:32s} {len(v):4d}  {m[0]:6.1f} {m[1]:5.1f}  {m[2]:7.1f}  {m[3]:7.1f}  {m[4]:9.1f}')


if __name__ == '__main__':
    main()
