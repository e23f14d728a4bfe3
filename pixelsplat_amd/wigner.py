"""Constants for the real Wigner-D matrices of degree <= 4 in e3nn's convention.

The reference rotates the SH coefficients of every Gaussian into world space with
`e3nn.o3.wigner_D(l, *matrix_to_angles(R))` (/root/reference/src/misc/sh_rotation.py:10-31),
where R = Y(alpha) X(beta) Y(gamma) and D^l = exp(alpha G_y) exp(beta G_x) exp(gamma G_y) with
G the so(3) generators on e3nn's real harmonics.  exp(t G_y) is a set of plane rotations by
m t (closed form), and G_x = P_l G_y P_l^T for the fixed matrix P_l = exp(-pi/2 G_z), so

    D^l(alpha, beta, gamma) = Z_l(alpha) P_l Z_l(beta) P_l^T Z_l(gamma).

The device kernel (csrc/gaussian_adapter.hip) evaluates that product per view; this module
only supplies the P_l, computed once in float64.
"""
from __future__ import annotations

import functools

import numpy as np

MAX_DEGREE = 4
BLOCK_OFFSETS = [0, 1, 10, 35, 84, 165]     # start of the (2l+1)^2 block of degree l


def _real_generators(l: int) -> np.ndarray:
    """so(3) generators (x, y, z in e3nn's axis naming -> indices 0, 1, 2) on the real basis
    obtained from the complex one by e3nn's unitary change of basis."""
    n = 2 * l + 1
    m = np.arange(-l, l + 1)
    up = np.zeros((n, n))
    dn = np.zeros((n, n))
    for i in range(n - 1):
        c = np.sqrt(l * (l + 1) - m[i] * (m[i] + 1))
        up[i + 1, i] = -c            # raising
        dn[i, i + 1] = c             # lowering
    jx = 0.5 * (up + dn).astype(complex)
    jz = np.diag(1j * m)
    jy = -0.5j * (up - dn)
    u = np.zeros((n, n), dtype=complex)
    s = 1 / np.sqrt(2)
    for k in range(1, l + 1):
        u[l - k, l + k] = s
        u[l - k, l - k] = -1j * s
        u[l + k, l + k] = (-1) ** k * s
        u[l + k, l - k] = 1j * (-1) ** k * s
    u[l, l] = 1
    u = (-1j) ** l * u
    gens = np.stack([u.conj().T @ g @ u for g in (jx, jz, jy)])
    assert np.abs(gens.imag).max() < 1e-12
    return gens.real


def _expm_skew(a: np.ndarray) -> np.ndarray:
    """exp of a real skew-symmetric matrix via its (unitary) eigen-decomposition."""
    w, v = np.linalg.eig(a)
    return (v @ np.diag(np.exp(w)) @ np.linalg.inv(v)).real


@functools.lru_cache(maxsize=None)
def conjugation_matrices() -> np.ndarray:
    """P_1 .. P_4 flattened row-major into one float64 vector of 164 entries (P_0 = 1)."""
    out = []
    for l in range(1, MAX_DEGREE + 1):
        g = _real_generators(l)
        p = _expm_skew(-0.5 * np.pi * g[2])
        assert np.abs(p @ g[1] @ p.T - g[0]).max() < 1e-10
        out.append(p.reshape(-1))
    return np.concatenate(out)
