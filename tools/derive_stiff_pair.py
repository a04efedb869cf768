"""Where the coefficients of the stability-optimised pair in include/simplyp_controller.h (SIMPLYP_STIFF_*) come from.

A reach far down a network relaxes to its quasi-steady flow at rate = cQ Qr**b_Q of up to several hundred per day; once the
transient that follows midnight has died away, Cash-Karp's steps there are bound by its real stability interval (|h x rate| <=
3.73), not by accuracy (tools/probe_c4_steps.py).  For those steps the kernel switches, per lane, to a second explicit 6-stage
pair made for the purpose -- same stage count, same sparsity of the weights as Cash-Karp (b2 = b5 = 0, e2 = 0), so the attempt
loop runs the same instructions with other constants:

  * order 4 (8 conditions), embedded order 3 (4 conditions);
  * |R(z)| <= 0.97 and |R_hat(z)| <= 1 on the real interval [-beta, 0]   (R, R_hat: stability functions of the two solutions);
  * internal stability: every stage polynomial |P_i(z)| <= pmax on that interval (Cash-Karp's own reach 250 at z = -8, which is
    why re-weighting ITS stages does not work: the nonlinear term of the flow equation no longer cancels);
  * all nodes in [0, 1]; small fifth-order error coefficients of the main solution.

Found by least squares from random starts (scipy `least_squares`, trust-region reflective, finite-difference Jacobian), then
projected onto the order conditions and the exact zeros to machine precision.  Deterministic for a given (--beta, --pmax, --seed,
--trial): the pair in the header is `--beta 9 --pmax 1.5 --seed 34 --trial 53`.  `--search N` runs N trials and lists them.

Usage: python tools/derive_stiff_pair.py [--beta 9 --pmax 1.5 --seed 34 --trial 53] [--search N] [--emit]
"""
import argparse
import os

os.environ.setdefault('OMP_NUM_THREADS', '1')          # 6 x 6 solves: threading only costs
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
import numpy as np
from scipy.optimize import least_squares

IDX = [(i, j) for i in range(6) for j in range(i)]
II, JJ = np.array([i for i, j in IDX]), np.array([j for i, j in IDX])
NAMES_A = ['A%d%d' % (i + 1, j + 1) for i, j in IDX]
E6, I6 = np.ones(6), np.eye(6)


def unpack(x):
    A = np.zeros((6, 6))
    A[II, JJ] = x[:15]
    return A, x[15:21], x[21:27]


def order4(A, b):
    """the 8 conditions of order 4 (rooted trees with <= 4 vertices), as residuals"""
    c = A.sum(1); Ac = A @ c; Ac2 = A @ (c * c); AAc = A @ Ac
    return np.array([b.sum() - 1, b @ c - 1 / 2, b @ (c * c) - 1 / 3, b @ Ac - 1 / 6,
                     b @ c ** 3 - 1 / 4, b @ (c * Ac) - 1 / 8, b @ Ac2 - 1 / 12, b @ AAc - 1 / 24])


def error5(A, b):
    """residuals of the 9 conditions of order 5: the principal error coefficients of a 4th-order solution"""
    c = A.sum(1); Ac = A @ c; Ac2 = A @ (c * c); AAc = A @ Ac
    return np.array([b @ c ** 4 - 1 / 5, b @ (c * c * Ac) - 1 / 10, b @ (c * Ac2) - 1 / 15, b @ (c * AAc) - 1 / 30, b @ (Ac * Ac) - 1 / 20,
                     b @ (A @ c ** 3) - 1 / 20, b @ (A @ (c * Ac)) - 1 / 40, b @ (A @ Ac2) - 1 / 60, b @ (A @ AAc) - 1 / 120])


def stage_polys(A, zs):
    M = I6[None] - zs[:, None, None] * A[None]
    return np.linalg.solve(M, np.broadcast_to(E6, (len(zs), 6))[..., None])[..., 0]       # [nz, 6]: Y_i = P_i(z) y for y' = lambda y


def search_residual(x, zs, pmax):
    A, b, bh = unpack(x); c = A.sum(1)
    P = stage_polys(A, zs)
    R, Rh = 1 + zs * (P @ b), 1 + zs * (P @ bh)
    return np.concatenate([order4(A, b) * 1e3, order4(A, bh)[:4] * 1e3,
                           np.maximum(0, np.abs(R) - 0.97) * 30, np.maximum(0, np.abs(Rh) - 1.0) * 30, np.maximum(0, np.abs(P) - pmax).ravel() * 10,
                           error5(A, b) * 10, 0.3 * order4(A, bh)[4:] * 10, [max(0, c.max() - 1.0) * 30, max(0, 0.05 - c[1:].min()) * 30], 1e-3 * x,
                           [max(0, 0.05 - np.abs(b - bh).max()) * 30], [b[1] * 1e3, b[4] * 1e3, bh[1] * 1e3]])


def start_point(rng):
    return np.concatenate([rng.uniform(-0.3, 0.6, 15), rng.uniform(0, 0.4, 6), rng.uniform(0, 0.4, 6)])


def polish(x):
    """onto the 12 order conditions with b2 = b5 = bhat2 = 0 exactly, moving the other 24 coefficients as little as possible"""
    x = x.copy()
    x[[16, 19, 22]] = 0.0
    free = np.array([i for i in range(27) if i not in (16, 19, 22)])

    def full(yf):
        y = x.copy(); y[free] = yf
        return y
    g = lambda yf: np.concatenate([order4(*unpack(full(yf))[:2]), order4(unpack(full(yf))[0], unpack(full(yf))[2])[:4], 1e-7 * (yf - x[free])])
    return full(least_squares(g, x[free], method='trf', xtol=3e-16, ftol=3e-16, gtol=3e-16, max_nfev=300).x)


def properties(x):
    A, b, bh = unpack(x)
    zs = np.linspace(-12, 0, 2401)
    P = stage_polys(A, zs)
    R, Rh = 1 + zs * (P @ b), 1 + zs * (P @ bh)
    bad = zs[np.abs(R) > 1 + 1e-9]
    beta = -bad.max() if len(bad) else 12.0
    ins = zs >= -beta
    return dict(beta=beta, max_stage_poly=float(np.abs(P[ins]).max()), max_Rhat=float(np.abs(Rh[ins]).max()),
                order4_residual=float(np.abs(order4(A, b)).max()), order3_residual=float(np.abs(order4(A, bh)[:4]).max()),
                error5_norm=float(np.linalg.norm(error5(A, b))), nodes=np.round(A.sum(1), 4), max_abs_a=float(np.abs(A).max()))


def one_trial(beta, pmax, rng):
    zs = np.linspace(-beta, 0, 121)[:-1]
    x0 = start_point(rng)
    r = least_squares(search_residual, x0, args=(zs, pmax), method='trf', max_nfev=600, xtol=1e-12, ftol=1e-12, gtol=1e-12)
    return r.x


def emit(x):
    A, b, bh = unpack(x)
    e = b - bh
    print('/* generated by tools/derive_stiff_pair.py --emit */')
    for (i, j), nm in zip(IDX, NAMES_A):
        print('#define SIMPLYP_STIFF_%s %.17g' % (nm, A[i, j]))
    for k in (0, 2, 3, 5):
        print('#define SIMPLYP_STIFF_B%d %.17g' % (k + 1, b[k]))
    for k in (0, 2, 3, 4, 5):
        print('#define SIMPLYP_STIFF_E%d %.17g' % (k + 1, e[k]))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--beta', type=float, default=9.0)
    ap.add_argument('--pmax', type=float, default=1.5)
    ap.add_argument('--seed', type=int, default=34)
    ap.add_argument('--trial', type=int, default=53)
    ap.add_argument('--search', type=int, default=0)
    ap.add_argument('--emit', action='store_true')
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    if args.search:
        for t in range(args.search):
            x = polish(one_trial(args.beta, args.pmax, rng))
            p = properties(x)
            print('trial %d: beta %.2f, stage polynomials <= %.2f, |R_hat| <= %.3f, order residuals %.0e / %.0e, error5 %.2e, max |a| %.2f, nodes %s'
                  % (t, p['beta'], p['max_stage_poly'], p['max_Rhat'], p['order4_residual'], p['order3_residual'], p['error5_norm'], p['max_abs_a'], p['nodes']), flush=True)
    else:
        for _ in range(args.trial):
            start_point(rng)                       # the starts of the earlier trials of this seed
        x = polish(one_trial(args.beta, args.pmax, rng))
        for k, v in properties(x).items():
            print('%-18s %s' % (k, v))
        if args.emit:
            emit(x)
