#!/usr/bin/env python
"""Per-shape table of the convolution launches of one step from bench.py's FIERY_BENCH_DUMP file(s).
  python tools/launches_table.py dump.json [other_dump.json]     (two files: side by side, for same-box A/B runs)"""
import json
import sys

PEAK = {'f32': 157.3, 'bf16': 2500.0}
# `work` in the dump is the direct form's flops; a Winograd F(2x2, 3x3) launch executes 16 / 36 of them on the matrix cores: its
# TFLOP/s and frac columns are on EXECUTED flops (the launch's time is what it is either way)
# (its split form: six bf16 MFMAs per fp32 product block, against the bf16 peak)
# (the split tile form: six per product, chained 1x1 parts included - an upper bound for the tails)
executed = lambda work, prec: (work * 16.0 / 36.0 * (6.0 if 'split' in str(prec) else 1.0) if 'winograd' in str(prec) else
                               work * 6.0 if 'split' in str(prec) else work)
peak_of = lambda prec: PEAK['bf16'] if (prec == 'bf16' or 'split' in str(prec)) else PEAK['f32']


def table(path):
    rows = {}
    for r in json.load(open(path)):
        if r['kind'] != 'conv_igemm':
            continue
        key = json.dumps(r['detail'])
        n, us, work = rows.get(key, (0, 0.0, 0.0))
        rows[key] = (n + 1, us + r['us'], work + r['work'])
    return rows


def main():
    tabs = [table(p) for p in sys.argv[1:3]]
    keys = sorted(tabs[0], key=lambda k: -tabs[0][k][1])
    tot = [sum(v[1] for v in t.values()) for t in tabs]
    work = sum(executed(v[2], json.loads(k)[-1]) for k, v in tabs[0].items())
    print('# detail = [kT, kH(=kW), stride, ?, cin_total, cout, images, Hout, Wout, precision]; TFLOP/s against the form\'s MFMA peak')
    hdr = '   n        us  share  TFLOP/s   frac' + ('        us  TFLOP/s   frac   ratio' if len(tabs) > 1 else '') + '  detail'
    print(hdr)
    for k in keys:
        n, us, w = tabs[0][k]
        prec = json.loads(k)[-1]
        tf = executed(w, prec) / us / 1e6
        line = f'{n:4d} {us:9.1f} {100 * us / tot[0]:5.1f}% {tf:8.1f} {tf / peak_of(prec):6.3f}'
        if len(tabs) > 1 and k in tabs[1]:
            _, us2, w2 = tabs[1][k]
            tf2 = executed(w2, prec) / us2 / 1e6
            line += f' {us2:9.1f} {tf2:8.1f} {tf2 / peak_of(prec):6.3f} {us2 / us:7.3f}'
        print(line + '  ' + k)
    line = f'{sum(v[0] for v in tabs[0].values()):4d} {tot[0]:9.1f} 100.0% {work / tot[0] / 1e6:8.1f} {work / tot[0] / 1e6 / 157.3:6.3f}'
    if len(tabs) > 1:
        line += f' {tot[1]:9.1f} {work / tot[1] / 1e6:8.1f} {work / tot[1] / 1e6 / 157.3:6.3f} {tot[1] / tot[0]:7.3f}'
    print(line + '  all (executed flops, frac against the fp32 peak)')


if __name__ == '__main__':
    main()
