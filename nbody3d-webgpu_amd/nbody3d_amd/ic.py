"""Initial conditions in the reference's host layout.

Every generator returns ``(bodies, vel)`` as float32 arrays of shape (N, 4):
``bodies[i] = (x, y, z, mass)``, ``vel[i] = (vx, vy, vz, 0)`` -- the packed
Float32Array layout the reference uploads (/root/reference nbody3d.js:49,
:66-68, :108-109, :123, :132).

The reference's only generator is ``generateGalaxy`` (nbody3d.js:51-133).  The
Plummer sphere and the uniform cube that BASELINE.json's configs name do not
exist upstream; they are defined here.
"""
import numpy as np


def plummer(n, seed=1, rcut=10.0):
    """Plummer sphere, Aarseth-Henon-Wielen (1974) sampling, N-body units
    (total mass 1, G = 1, virial radius 1), radius cut at ``rcut`` scale radii,
    centre of mass and total momentum removed.  BASELINE.json configs 1, 3, 5."""
    rng = np.random.default_rng(seed)
    # radii: r = (X^(-2/3) - 1)^(-1/2), reject r > rcut
    r = np.empty(n)
    filled = 0
    while filled < n:
        x1 = rng.random(n - filled)
        x1 = x1[x1 > 1e-10]
        rr = 1.0 / np.sqrt(x1 ** (-2.0 / 3.0) - 1.0)
        rr = rr[rr <= rcut]
        r[filled:filled + rr.size] = rr
        filled += rr.size
    pos = _isotropic(rng, n) * r[:, None]
    # speeds: q = v / v_esc with density g(q) = q^2 (1 - q^2)^(7/2), rejection
    q = np.empty(n)
    filled = 0
    while filled < n:
        m = n - filled
        x4 = rng.random(m)
        x5 = rng.random(m)
        ok = 0.1 * x5 < x4 * x4 * (1.0 - x4 * x4) ** 3.5
        qq = x4[ok]
        q[filled:filled + qq.size] = qq
        filled += qq.size
    vesc = np.sqrt(2.0) * (1.0 + r * r) ** -0.25
    v = _isotropic(rng, n) * (q * vesc)[:, None]
    # scale radius a = 3*pi/16 gives virial radius 1 at M = G = 1
    a = 3.0 * np.pi / 16.0
    pos *= a
    v /= np.sqrt(a)
    pos -= pos.mean(axis=0)
    v -= v.mean(axis=0)
    bodies = np.zeros((n, 4), np.float32)
    vel = np.zeros((n, 4), np.float32)
    bodies[:, :3] = pos
    bodies[:, 3] = 1.0 / n
    vel[:, :3] = v
    return bodies, vel


def uniform_cube(n, seed=2):
    """Positions uniform in [-1,1)^3, masses uniform in [0.5,1.5)/N, at rest.
    BASELINE.json config 2."""
    rng = np.random.default_rng(seed)
    bodies = np.zeros((n, 4), np.float32)
    bodies[:, :3] = rng.random((n, 3)) * 2.0 - 1.0
    bodies[:, 3] = (rng.random(n) + 0.5) / n
    return bodies, np.zeros((n, 4), np.float32)


def _isotropic(rng, n):
    z = rng.random(n) * 2.0 - 1.0
    phi = rng.random(n) * 2.0 * np.pi
    s = np.sqrt(1.0 - z * z)
    return np.stack([s * np.cos(phi), s * np.sin(phi), z], axis=1)
