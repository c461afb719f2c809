import numpy as np
from scipy.special import ndtr
from scipy.optimize import least_squares
for X in (4.5, 5.5):
  x = np.linspace(1e-3, X, 8001)
  phi = ndtr(x); g = x * phi
  s = x * x
  for deg in (3, 4, 5, 6):
    A = np.vander(s, deg + 1, increasing=True)
    def res(c, w):
        y = x * (A @ c)
        return (x / (1 + np.exp(-y)) - g) * w
    c0 = np.zeros(deg + 1); c0[0] = 1.5957691; c0[1] = 0.0713548
    w = np.ones_like(x)
    c = least_squares(res, c0, args=(w,), x_scale=np.abs(c0) + 10.0 ** (-2 * np.arange(deg + 1) - 1), xtol=1e-15, ftol=1e-15, gtol=1e-15).x
    for it in range(40):   # Lawson
        e = np.abs(res(c, 1.0))
        w = w * (e / e.mean()) ** 0.5; w /= w.mean()
        c = least_squares(res, c, args=(w,), xtol=1e-15, ftol=1e-15, gtol=1e-15).x
    e = res(c, 1.0)
    print(X, deg, "max abs err %.3e at %.3f" % (np.abs(e).max(), x[np.abs(e).argmax()]), repr(c))
