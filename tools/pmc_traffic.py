#!/usr/bin/env python
"""Folds the two PMC passes (tools/pmc_dump.py output of `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate
runs of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph`) into HBM traffic per launch.
Correction (MI355X_MICROARCH.md, HBM section): both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of 16-byte-per-lane loads at 64 B, so it is doubled for these kernels (all their bulk reads are dwordx4);
WRITE_SIZE is taken as is (k_lift_geometry writes exactly its 52.25 MB: 51030 KB counted).
  usage: python tools/pmc_traffic.py FETCH.txt WRITE.txt out.json"""
import json
import re
import sys


def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r'(fiery::\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)\s+total=\s*([\d.]+)', line)
        if m:
            rows[m.group(1).strip()] = (int(m.group(3)), float(m.group(5)))
    return rows


def main():
    fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
    out = {'_note': 'bytes per launch; traffic = 2 * FETCH_SIZE + WRITE_SIZE (KB -> B); see tools/pmc_traffic.py'}
    groups = {'k_conv_igemm (all tile shapes)': [k for k in fetch if 'k_conv_igemm' in k],
              'convolutions (all forms)': [k for k in fetch if 'k_conv_igemm' in k or 'k_conv_winograd' in k],
              'k_voxel_pool': [k for k in fetch if 'k_voxel_pool' in k]}
    for k in sorted(fetch):
        groups[k] = [k]
    for name, keys in groups.items():
        nf = sum(fetch[k][0] for k in keys)
        nw = sum(write[k][0] for k in keys if k in write)
        if not nf or not nw:
            continue
        f = 2.0 * 1024 * sum(fetch[k][1] for k in keys) / nf
        w = 1024.0 * sum(write[k][1] for k in keys if k in write) / nw
        out[name] = {'launches_fetch_pass': nf, 'launches_write_pass': nw, 'read_bytes': round(f), 'write_bytes': round(w),
                     'traffic_bytes': round(f + w)}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k in ('k_conv_igemm (all tile shapes)', 'k_voxel_pool'):
        print(k, out.get(k))


if __name__ == '__main__':
    main()
