"""Coefficients for polynomial smoothers (setup-time constants).  Same values as
/root/reference/pyamg/relaxation/chebyshev.py:12-56 (pinned by tests/test_host_api.py and the
reference-built hierarchies under tests/golden)."""
import numpy as np

__all__ = ["chebyshev_polynomial_coefficients"]


def chebyshev_polynomial_coefficients(a, b, degree):
    """Descending coefficients of the degree-`degree` polynomial p with p(0) = 1 that is smallest
    in the max-norm on [a, b]: the Chebyshev polynomial of the interval, normalised at the origin.

    Its zeros are the Chebyshev nodes cos(pi (k + 1/2) / degree) carried from [-1, 1] to [a, b];
    the monic product over the zeros is expanded one linear factor at a time and divided by its
    constant term (= its value at 0)."""
    if not (0 < a < b):
        raise ValueError("invalid interval [%s,%s]" % (a, b))
    nodes = np.cos(np.pi * (np.arange(degree) + 0.5) / degree)
    zeros = 0.5 * (b - a) * (1 + nodes) + a
    monic = np.ones(1)
    for z in zeros:
        monic = np.convolve(monic, np.array([1.0, -z]))
    return monic / monic[-1]
