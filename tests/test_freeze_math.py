"""The algebra behind the frozen pairs (slam_sensor_fusion_amd/csrc/sf_icp.hip: mom_term / frozen_record), restated in numpy and
checked against the direct sums: with the pairs (x, p, n) fixed and y = R x + t, the P2PLANE record of an iteration

    count,  sum r^2,  sum J J^T (21),  sum J r (6),  sum |y - p|^2        r = n.(y - p),  J = [y x n; n]

is a polynomial of degree <= 2 in (R, t) whose coefficients are 96 moment sums over the pairs.  The layout below is the
kernel's (k6(a, b): 00 01 02 11 12 22):

    [0,36)  n_c n_f x_d x_e   (6 k6(c,f) + k6(d,e))      [36,54) n_c n_f x_d (36 + 3 k6 + d)     [54,60) n_c n_f
    [60,69) c n_a x_d (60 + 3 a + d)   [69,72) c n_a   72 c^2   73 count         (c = n.p)
    [74,80) x_d x_e   [80,83) x_d   [83,92) p_a x_d (83 + 3 a + d)   [92,95) p_a   95 |p|^2

No reference counterpart: the reference evaluates every pair in every iteration (localization/src/icp_point_to_point.cpp:64-69;
the point-to-plane form is north_star's).  CPU only: the GPU side is tests/test_gpu_freeze.py."""
import numpy as np

K6 = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]


def k6(a, b):
    a, b = min(a, b), max(a, b)
    return K6.index((a, b))


def moments(x, n, p):
    c = (n * p).sum(1)
    m = np.zeros(96)
    for cf, (a, b) in enumerate(K6):
        nn = n[:, a] * n[:, b]
        for de, (d, e) in enumerate(K6):
            m[6 * cf + de] = (nn * x[:, d] * x[:, e]).sum()
        for d in range(3):
            m[36 + 3 * cf + d] = (nn * x[:, d]).sum()
        m[54 + cf] = nn.sum()
    for a in range(3):
        for d in range(3):
            m[60 + 3 * a + d] = (c * n[:, a] * x[:, d]).sum()
            m[83 + 3 * a + d] = (p[:, a] * x[:, d]).sum()
        m[69 + a] = (c * n[:, a]).sum()
        m[80 + a] = x[:, a].sum()
        m[92 + a] = p[:, a].sum()
    m[72] = (c * c).sum()
    m[73] = len(x)
    for de, (d, e) in enumerate(K6):
        m[74 + de] = (x[:, d] * x[:, e]).sum()
    m[95] = (p * p).sum()
    return m


def record_from_moments(m, R, t):
    """frozen_record of sf_icp.hip: the 30 sums at the pose (R, t) from the moments."""
    YY = np.zeros((6, 3, 3))   # sum n_c n_f y_b y_e
    YN = np.zeros((6, 3))      # sum n_c n_f y_b
    CY = np.zeros((3, 3))      # sum c n_a y_b
    for cf in range(6):
        M4 = np.zeros((3, 3))
        for d in range(3):
            for e in range(3):
                M4[d, e] = m[6 * cf + k6(d, e)]
        M3 = m[36 + 3 * cf: 39 + 3 * cf]
        YY[cf] = R @ M4 @ R.T + np.outer(R @ M3, t) + np.outer(t, R @ M3) + np.outer(t, t) * m[54 + cf]
        YN[cf] = R @ M3 + t * m[54 + cf]
    for a in range(3):
        CY[a] = R @ m[60 + 3 * a: 63 + 3 * a] + t * m[69 + a]
    yy = lambda c, f, b, e: YY[k6(c, f)][b, e]
    A = np.zeros((6, 6))
    rhs = np.zeros(6)
    for a in range(3):
        b1, c1 = (a + 1) % 3, (a + 2) % 3
        for a2 in range(3):
            b2, c2 = (a2 + 1) % 3, (a2 + 2) % 3
            A[a, a2] = yy(c1, c2, b1, b2) - yy(c1, b2, b1, c2) - yy(b1, c2, c1, b2) + yy(b1, b2, c1, c2)
        for f in range(3):
            A[a, 3 + f] = A[3 + f, a] = YN[k6(c1, f)][b1] - YN[k6(b1, f)][c1]
        rhs[a] = sum(yy(c1, e, b1, e) - yy(b1, e, c1, e) for e in range(3)) - CY[c1][b1] + CY[b1][c1]
    for c in range(3):
        for f in range(3):
            A[3 + c, 3 + f] = m[54 + k6(c, f)]
        rhs[3 + c] = sum(YN[k6(c, e)][e] for e in range(3)) - m[69 + c]
    r2 = m[72] + sum(yy(c, f, c, f) for c in range(3) for f in range(3)) - 2.0 * sum(CY[c][c] for c in range(3))
    XX = np.zeros((3, 3))
    for d in range(3):
        for e in range(3):
            XX[d, e] = m[74 + k6(d, e)]
    X1, PX, P1 = m[80:83], m[83:92].reshape(3, 3), m[92:95]
    sd = np.trace(R @ XX @ R.T) + 2.0 * t @ (R @ X1) + m[73] * t @ t - 2.0 * (np.trace(R @ PX.T) + t @ P1) + m[95]
    return m[73], r2, A, rhs, sd


def record_direct(x, n, p, R, t):
    y = x @ R.T + t
    r = (n * (y - p)).sum(1)
    J = np.concatenate([np.cross(y, n), n], axis=1)
    return len(x), (r * r).sum(), J.T @ J, J.T @ r, ((y - p) ** 2).sum()


def rigid(rng, angle, shift):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(rng.normal(0, angle, 3)).as_matrix(), rng.normal(0, shift, 3)


def test_record_is_a_polynomial_of_the_moments():
    rng = np.random.default_rng(3)
    for trial in range(6):
        N = int(rng.integers(5, 4000))
        x = rng.uniform(-50, 50, (N, 3)) * [1, 1, 0.1]
        n = rng.normal(0, 1, (N, 3))
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        p = x + rng.normal(0, 0.02, (N, 3))
        m = moments(x, n, p)
        for angle, shift in ((0.0, 0.0), (1e-4, 1e-3), (0.3, 2.0)):       # the identity, an ICP step, anything
            R, t = rigid(rng, angle, shift)
            got, want = record_from_moments(m, R, t), record_direct(x, n, p, R, t)
            for g, w in zip(got, want):
                scale = max(np.abs(np.asarray(w, dtype=np.float64)).max(), 1.0) * (1.0 + N * 2500.0)   # terms of size |x|^2 cancel
                assert np.abs(np.asarray(g) - np.asarray(w)).max() <= 1e-13 * scale, (trial, angle)


def test_moment_layout_is_the_kernels():
    """The indices the kernel's mom_term<I> uses, spelt out for one pair."""
    x, n, p = np.array([[2.0, 3.0, 5.0]]), np.array([[0.5, -0.25, 0.125]]), np.array([[7.0, 11.0, 13.0]])
    m = moments(x, n, p)
    c = float((n * p).sum())
    assert m[0] == n[0, 0] ** 2 * x[0, 0] ** 2 and m[6 * 1 + 4] == (n[0, 0] * n[0, 1]) * (x[0, 1] * x[0, 2])      # [k6(0,1)][k6(1,2)]
    assert m[36 + 3 * 5 + 1] == n[0, 2] ** 2 * x[0, 1] and m[54 + 2] == n[0, 0] * n[0, 2]
    assert m[60 + 3 * 1 + 2] == c * n[0, 1] * x[0, 2] and m[69 + 2] == c * n[0, 2] and m[72] == c * c and m[73] == 1
    assert m[74 + 4] == x[0, 1] * x[0, 2] and m[80 + 1] == x[0, 1] and m[83 + 3 * 2 + 0] == p[0, 2] * x[0, 0] and m[92] == p[0, 0] and m[95] == (p * p).sum()
