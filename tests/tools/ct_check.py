"""Development check: a restatement of the Clough-Tocher evaluation (SciPy's
CloughTocher2DInterpolator: cubic C1 interpolant on a 2-D Delaunay triangulation with
vertex gradients) against SciPy itself.  SciPy is a third-party dependency of the
reference (optics/_ShapeMesh.py:226-229), absent from /root/reference; its algorithm is
restated from Alfeld / Farin as implemented there (affine-invariant edge directions
through the neighbour's centroid)."""
import numpy as np
from scipy.spatial import Delaunay
from scipy.interpolate import CloughTocher2DInterpolator


def barycentric(transform, isimplex, x):
    T = transform[isimplex]
    c = np.zeros(3)
    c[2] = 1.0
    for i in range(2):
        c[i] = 0.0
        for j in range(2):
            c[i] += T[i, j] * (x[j] - T[2, j])
        c[2] -= c[i]
    return c


def ct_single(points, simplices, neighbors, transform, isimplex, b, f, df):
    v = simplices[isimplex]
    e12x = points[v[1], 0] - points[v[0], 0]; e12y = points[v[1], 1] - points[v[0], 1]
    e23x = points[v[2], 0] - points[v[1], 0]; e23y = points[v[2], 1] - points[v[1], 1]
    e31x = points[v[0], 0] - points[v[2], 0]; e31y = points[v[0], 1] - points[v[2], 1]
    f1, f2, f3 = f
    df12 = +(df[0, 0] * e12x + df[0, 1] * e12y)
    df21 = -(df[1, 0] * e12x + df[1, 1] * e12y)
    df23 = +(df[1, 0] * e23x + df[1, 1] * e23y)
    df32 = -(df[2, 0] * e23x + df[2, 1] * e23y)
    df31 = +(df[2, 0] * e31x + df[2, 1] * e31y)
    df13 = -(df[0, 0] * e31x + df[0, 1] * e31y)
    c3000 = f1
    c2100 = (df12 + 3 * c3000) / 3
    c2010 = (df13 + 3 * c3000) / 3
    c0300 = f2
    c1200 = (df21 + 3 * c0300) / 3
    c0210 = (df23 + 3 * c0300) / 3
    c0030 = f3
    c1020 = (df31 + 3 * c0030) / 3
    c0120 = (df32 + 3 * c0030) / 3
    c2001 = (c2100 + c2010 + c3000) / 3
    c0201 = (c1200 + c0300 + c0210) / 3
    c0021 = (c1020 + c0120 + c0030) / 3
    g = np.zeros(3)
    for k in range(3):
        itri = neighbors[isimplex, k]
        if itri == -1:
            g[k] = -1. / 2
            continue
        w = simplices[itri]
        y = np.array([(points[w[0], 0] + points[w[1], 0] + points[w[2], 0]) / 3,
                      (points[w[0], 1] + points[w[1], 1] + points[w[2], 1]) / 3])
        c = barycentric(transform, isimplex, y)
        if k == 0:
            g[k] = (2 * c[2] + c[1] - 1) / (2 - 3 * c[2] - 3 * c[1])
        elif k == 1:
            g[k] = (2 * c[0] + c[2] - 1) / (2 - 3 * c[0] - 3 * c[2])
        else:
            g[k] = (2 * c[1] + c[0] - 1) / (2 - 3 * c[1] - 3 * c[0])
    c0111 = (g[0] * (-c0300 + 3 * c0210 - 3 * c0120 + c0030) + (-c0300 + 2 * c0210 - c0120 + c0021 + c0201)) / 2
    c1011 = (g[1] * (-c0030 + 3 * c1020 - 3 * c2010 + c3000) + (-c0030 + 2 * c1020 - c2010 + c2001 + c0021)) / 2
    c1101 = (g[2] * (-c3000 + 3 * c2100 - 3 * c1200 + c0300) + (-c3000 + 2 * c2100 - c1200 + c2001 + c0201)) / 2
    c1002 = (c1101 + c1011 + c2001) / 3
    c0102 = (c1101 + c0111 + c0201) / 3
    c0012 = (c1011 + c0111 + c0021) / 3
    c0003 = (c1002 + c0102 + c0012) / 3
    minval = min(b)
    b1, b2, b3, b4 = b[0] - minval, b[1] - minval, b[2] - minval, 3 * minval
    w = (b1**3*c3000 + 3*b1**2*b2*c2100 + 3*b1**2*b3*c2010 + 3*b1**2*b4*c2001 + 3*b1*b2**2*c1200 +
         6*b1*b2*b4*c1101 + 3*b1*b3**2*c1020 + 6*b1*b3*b4*c1011 + 3*b1*b4**2*c1002 + b2**3*c0300 +
         3*b2**2*b3*c0210 + 3*b2**2*b4*c0201 + 3*b2*b3**2*c0120 + 6*b2*b3*b4*c0111 + 3*b2*b4**2*c0102 +
         b3**3*c0030 + 3*b3**2*b4*c0021 + 3*b3*b4**2*c0012 + b4**3*c0003)
    return w


if __name__ == '__main__':
    rng = np.random.RandomState(0)
    pts = rng.uniform(-1, 1, (200, 2))
    vals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + pts[:, 0] ** 2
    tri = Delaunay(pts)
    ct = CloughTocher2DInterpolator(tri, vals)
    grad = ct.grad[:, 0, :]
    xs = rng.uniform(-0.9, 0.9, (3000, 2))
    ref = ct(xs[:, 0], xs[:, 1])
    isimp = tri.find_simplex(xs)
    worst = 0.0
    for x, r, s in zip(xs, ref, isimp):
        if s < 0 or np.isnan(r):
            continue
        b = barycentric(tri.transform, s, x)
        v = tri.simplices[s]
        w = ct_single(tri.points, tri.simplices, tri.neighbors, tri.transform, s, b, vals[v], grad[v])
        worst = max(worst, abs(w - r))
    print('max |mine - scipy| =', worst)
