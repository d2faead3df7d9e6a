"""Coefficients for polynomial smoothers (setup-time constants) -- follows
/root/reference/pyamg/relaxation/chebyshev.py:12-56."""
import numpy as np

__all__ = ["chebyshev_polynomial_coefficients"]


def chebyshev_polynomial_coefficients(a, b, degree):
    """Coefficients (descending order) of the Chebyshev polynomial C(t) of
    minimum magnitude on [a, b] with C(0) = 1."""
    if a >= b or a <= 0:
        raise ValueError("invalid interval [%s,%s]" % (a, b))
    std_roots = np.cos(np.pi * (np.arange(degree) + 0.5) / degree)
    scaled_roots = 0.5 * (b - a) * (1 + std_roots) + a
    scaled_poly = np.poly(scaled_roots)
    scaled_poly /= np.polyval(scaled_poly, 0)
    return scaled_poly
