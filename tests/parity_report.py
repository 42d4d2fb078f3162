"""Achieved-error ledger of the parity tests.

Every float comparison of the GPU tier records what it measured, not only whether it passed: the max-abs error
against the reference-produced value (fixture or live oracle), against the float64 evaluation where one exists,
the tensor's scale, the literal 1e-4 of BASELINE.json's north_star and the scale-relative bound the assertion uses.
`tests/conftest.py` prints the table at the end of the session and writes it to `gpurun_out/parity_errors.json`
(copied to `profiles/` per round), so the relaxation `1e-4 * max(1, |ref|_inf)` is auditable per output key.
"""
import json
import os

ROWS = []
LITERAL_TOL = 1e-4


def record(test, key, err_ref, scale, err_exact=None, ref_noise=None, bound=None, note=''):
    """err_ref: max |ours - reference|; err_exact: max |ours - float64 evaluation| (None when there is none);
    ref_noise: max |reference - float64 evaluation| (how far the reference itself is from what it approximates)."""
    ROWS.append(dict(test=test, key=key, err_vs_reference=float(err_ref), scale=float(scale),
                     err_vs_float64=None if err_exact is None else float(err_exact),
                     reference_vs_float64=None if ref_noise is None else float(ref_noise),
                     literal_tol=LITERAL_TOL, within_literal=bool(err_ref <= LITERAL_TOL),
                     asserted_bound=float(bound) if bound is not None else LITERAL_TOL * max(1.0, float(scale)), note=note))


def table():
    if not ROWS:
        return ''
    head = f'{"test":58s} {"key":20s} {"|ours-ref|":>11s} {"|ours-f64|":>11s} {"|ref-f64|":>10s} {"scale":>8s} {"<=1e-4":>6s} {"bound":>9s}'
    lines = [head, '-' * len(head)]
    fmt = lambda v: '       -' if v is None else f'{v:11.3e}'
    for r in ROWS:
        lines.append(f'{r["test"][:58]:58s} {r["key"][:20]:20s} {r["err_vs_reference"]:11.3e} {fmt(r["err_vs_float64"])} '
                     f'{fmt(r["reference_vs_float64"])[1:]} {r["scale"]:8.3f} {"yes" if r["within_literal"] else "NO":>6s} '
                     f'{r["asserted_bound"]:9.2e}')
    worst = max(ROWS, key=lambda r: r['err_vs_reference'])
    n_lit = sum(r['within_literal'] for r in ROWS)
    lines.append(f'{len(ROWS)} float comparisons; {n_lit} within the literal 1e-4; worst |ours-ref| = '
                 f'{worst["err_vs_reference"]:.3e} ({worst["test"]} / {worst["key"]}, scale {worst["scale"]:.2f})')
    return '\n'.join(lines)


def dump(path):
    if not ROWS:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(ROWS, f, indent=1)
