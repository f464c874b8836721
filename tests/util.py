"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import numpy as np

NCOMP = 21
G2 = 2     # guard cells for order-2 shapes (fields/Fields.cpp:63-64)


def thermal_sheet(nx, ny, lo, hi, ppc=2, seed=12345, u_std=0.1, jitter=1.0):
    """SURVEY 8(d) micro-benchmark sheet: lattice + uniform jitter, thermal momenta.

    Returns real (11, n) float64 in PlasmaIdx order, valid (n,) int32, ion_lev (n,) int32.
    """
    rng = np.random.default_rng(seed)
    dx = (hi[0] - lo[0]) / nx
    dy = (hi[1] - lo[1]) / ny
    xs, ys = [], []
    for ip in range(ppc * ppc):       # ppc index outermost (PlasmaParticleContainerInit.cpp:192)
        ixp, iyp = ip % ppc, ip // ppc
        ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
        xs.append((lo[0] + (ii + (0.5 + ixp) / ppc) * dx).ravel())
        ys.append((lo[1] + (jj + (0.5 + iyp) / ppc) * dy).ravel())
    x = np.concatenate(xs)
    y = np.concatenate(ys)
    n = x.size
    x = x + jitter * (rng.random(n) - 0.5) * dx
    y = y + jitter * (rng.random(n) - 0.5) * dy
    eps = 1e-9
    x = np.clip(x, lo[0] + eps, hi[0] - eps)
    y = np.clip(y, lo[1] + eps, hi[1] - eps)
    ux = rng.normal(0.0, u_std, n)
    uy = rng.normal(0.0, u_std, n)
    uz = rng.normal(0.0, u_std, n)
    psi = np.sqrt(1.0 + ux * ux + uy * uy + uz * uz) - uz
    w = np.full(n, 1.0 / (ppc * ppc)) * (0.5 + rng.random(n))
    real = np.empty((11, n))
    real[0], real[1], real[2], real[3], real[4], real[5] = x, y, w, ux, uy, psi
    real[6], real[7] = x, y
    real[8] = ux + rng.normal(0.0, 0.01, n)
    real[9] = uy + rng.normal(0.0, 0.01, n)
    real[10] = psi * (1.0 + rng.normal(0.0, 0.01, n))
    return real, np.ones(n, dtype=np.int32), np.zeros(n, dtype=np.int32)


def smooth_slab(nx, ny, g, ncomp=NCOMP, seed=7, amp=0.3):
    """Random smooth fields in every component (a few Fourier modes), guards included."""
    rng = np.random.default_rng(seed)
    jj, ii = np.meshgrid(np.arange(-g, ny + g), np.arange(-g, nx + g), indexing="ij")
    out = np.zeros((ncomp, ny + 2 * g, nx + 2 * g))
    for n in range(ncomp):
        for _ in range(4):
            kx, ky = rng.integers(1, 5, 2)
            ph = rng.random(2) * 2 * np.pi
            out[n] += amp * rng.normal() * np.sin(2 * np.pi * kx * ii / nx + ph[0]) * np.cos(2 * np.pi * ky * jj / ny + ph[1])
    return out


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale
