"""The Thompson-tau table (bronko_amd/host/tcrit_table.inc = oracle/tcrit_table.inc, both written by oracle/gen_tcrit.py) is the one
piece of arithmetic product and checker share: no parity test can catch a wrong quantile there.  This is an independent
spot check: the Student-t quantile of call.rs:924-925 (statrs StudentsT::new(0, 1, n - 2).inverse_cdf(1 - 0.001 / n)) recomputed at
50 digits WITHOUT the incomplete beta function the generator uses -- closed forms for 1 and 2 degrees of freedom, numerical
quadrature of the density otherwise -- and every tabulated double must be the nearest double to it: all 298 entries, n = 3 .. 300
(round 6; five of them until then)."""
import math
import os
import re

import pytest

mp = pytest.importorskip("mpmath")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(path):
    vals = {}
    for line in open(path):
        m = re.match(r"\s*(0x[0-9a-fA-F.]+p[+-]?\d+),\s*/\* n=(\d+) \*/", line)
        if m:
            vals[int(m.group(2))] = float.fromhex(m.group(1))
    return vals


def _cdf(t, df):
    """P(T <= t), t >= 0, by quadrature of the density (no special function but gamma)."""
    c = mp.gamma((df + 1) / mp.mpf(2)) / (mp.sqrt(df * mp.pi) * mp.gamma(df / mp.mpf(2)))
    dens = lambda x: c * (1 + x * x / df) ** (-(df + 1) / mp.mpf(2))
    # the tail instead of the body (p is within 1e-3 / n of 1: no cancellation), split where the integrand's scale changes
    pts = [t, 2 * t, 8 * t, 64 * t, 1024 * t, mp.inf]
    return 1 - mp.quad(dens, pts), dens(t)


def test_both_tables_are_the_same_file_and_complete():
    a = _table(os.path.join(ROOT, "oracle", "tcrit_table.inc"))
    b = _table(os.path.join(ROOT, "bronko_amd", "host", "tcrit_table.inc"))
    assert a == b and sorted(a) == list(range(3, 301))


@pytest.mark.parametrize("n", list(range(3, 301)))
def test_tabulated_quantile_is_the_nearest_double(n):
    mp.mp.dps = 50
    tab = _table(os.path.join(ROOT, "oracle", "tcrit_table.inc"))
    v = tab[n]
    p = mp.mpf(1.0 - 0.001 / float(n))        # the f64 argument exactly as call.rs:925 forms it
    df = n - 2
    if df == 1:                               # Cauchy: t = tan(pi (p - 1/2))
        exact = mp.tan(mp.pi * (p - mp.mpf(1) / 2))
    elif df == 2:                             # t = (2 p - 1) / sqrt(2 p (1 - p))
        exact = (2 * p - 1) / mp.sqrt(2 * p * (1 - p))
    else:                                     # one Newton step from the tabulated value: F(v) by quadrature, F' = the density
        F, f = _cdf(mp.mpf(v), mp.mpf(df))
        exact = mp.mpf(v) - (F - p) / f
        F2, f2 = _cdf(exact, mp.mpf(df))      # ... and it has converged far below an ulp
        assert abs((F2 - p) / f2) < mp.mpf(2) ** -80 * exact
    ulp = math.ulp(v)
    assert abs(mp.mpf(v) - exact) <= mp.mpf(ulp) / 2 * (1 + mp.mpf(10) ** -12), (n, v, exact)
    # the values SURVEY A.5 quotes from scipy for tau = t (n - 1) / (sqrt(n) sqrt(n - 2 + t^2))
    tau = v * (n - 1) / (math.sqrt(n) * math.sqrt(n - 2 + v * v))
    want = {3: 1.15469990524, 10: 2.6059340379, 100: 4.08396599115, 300: 4.43207983726}
    if n in want:
        assert abs(tau - want[n]) < 1e-10
