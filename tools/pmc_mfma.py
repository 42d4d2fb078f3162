#!/usr/bin/env python
"""Matrix-pipe utilisation per kernel from a PMC pass (tools/pmc_dump.py output of
`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace`).
On MI355X SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (a launch of N fp32 32x32x2 MFMAs reads 64 N)
and GRBM_GUI_ACTIVE over its 8 XCDs (GRBM_GUI_ACTIVE / 8 / launch duration = the shader clock), so
   MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8).
usage: python tools/pmc_mfma.py pmc_mfma.txt > out.txt"""
import collections
import re
import sys


def main():
    rows = collections.defaultdict(dict)
    for line in open(sys.argv[1]):
        m = re.match(r'(\S.*?)\s+(SQ_\w+|GRBM_\w+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)\s+total=\s*([\d.]+)', line)
        if m:
            rows[m.group(1).strip()][m.group(2)] = (int(m.group(3)), float(m.group(5)))
    print('# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); totals over all launches of the pass')
    print(f'{"kernel":62s} {"launches":>8s} {"mfma busy":>16s} {"gui active/8":>14s} {"MFMA util":>9s} {"VALU instr per MFMA":>20s}')
    tot_busy = tot_act = 0.0
    for name, c in sorted(rows.items()):
        if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c or c['SQ_VALU_MFMA_BUSY_CYCLES'][1] == 0:
            continue
        busy, act = c['SQ_VALU_MFMA_BUSY_CYCLES'][1], c['GRBM_GUI_ACTIVE'][1] / 8.0
        valu = c.get('SQ_INSTS_VALU', (0, 0.0))[1]
        n_mfma = busy / 64.0                       # fp32 32x32x2: 64 cycles each
        print(f'{name[:62]:62s} {c["GRBM_GUI_ACTIVE"][0]:8d} {busy:16.0f} {act:14.0f} {busy / (1024 * act):9.1%} {(valu - n_mfma) / n_mfma:20.2f}')
        if 'k_conv_igemm' in name:
            tot_busy += busy
            tot_act += act
    if tot_act:
        print(f'{"k_conv_igemm, all tile shapes":62s} {"":8s} {tot_busy:16.0f} {tot_act:14.0f} {tot_busy / (1024 * tot_act):9.1%}')


if __name__ == '__main__':
    main()
