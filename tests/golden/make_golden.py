#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

Source of truth: oracle/nb_oracle.c (the CPU restatement of the reference's
force+integrate pass, /root/reference nbody3d.js:232-291) applied to initial
conditions from nbody3d_amd.ic.  The reference itself holds no fixtures and its
WGSL cannot be executed in this image, so these vectors pin
the oracle against regressions and give the HIP engine and the Node wrapper a
shared, bit-stable target; they do not pin the oracle to a browser.

Run from the repo root:  python tests/golden/make_golden.py
Files are raw little-endian arrays (.f32 / .f64) + manifest.json.
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nbody3d-webgpu_amd"))

from oracle import oracle  # noqa: E402
from nbody3d_amd import ic  # noqa: E402


def save(name, arr):
    arr = np.ascontiguousarray(arr)
    ext = {np.dtype("float32"): ".f32", np.dtype("float64"): ".f64"}[arr.dtype]
    arr.astype(arr.dtype.newbyteorder("<")).tofile(os.path.join(HERE, name + ext))


def energy_report(b0, v0, G, dt, nsteps):
    """|dE/E0| of the fp32 and fp64 oracles.  vel after call k is synchronised
    with the positions BEFORE call k (SURVEY.md §8(c) energy note)."""
    out = {}
    ke0, pe0, _ = oracle.energy(b0, v0, G)
    e0 = ke0 + pe0
    for tag, run in (("f32", oracle.run_f32), ("f64", oracle.run_f64)):
        bprev, vprev, aprev = run(b0, v0, None, dt, G, nsteps - 1)
        bk, vk, ak = run(bprev, vprev, aprev, dt, G, 1)
        ke, _, _ = oracle.energy(bk, vk, G)          # KE from vel after call nsteps
        _, pe, _ = oracle.energy(bprev, vprev, G)    # PE from positions before it
        out[tag] = abs((ke + pe - e0) / e0)
    return e0, out


def case(manifest, name, bodies, vel, dt, G, checkpoints, with_f64=True):
    n = bodies.shape[0]
    save(name + "_bodies0", bodies)
    save(name + "_vel0", vel)
    entry = {"n": n, "dt": dt, "G": G, "eps2": oracle.EPS2, "checkpoints": checkpoints}
    b, v, a = bodies, vel, None
    done = 0
    for k in checkpoints:
        b, v, a = oracle.run_f32(b, v, a, dt, G, k - done)
        done = k
        save("%s_s%d_bodies" % (name, k), b)
        save("%s_s%d_vel" % (name, k), v)
        save("%s_s%d_accel" % (name, k), a)
    if with_f64:
        kmax = checkpoints[-1]
        b64, v64, a64 = oracle.run_f64(bodies, vel, None, dt, G, kmax)
        save("%s_s%d_bodies" % (name, kmax), b64)
        rscale = float(np.sqrt((bodies[:, :3].astype(np.float64) ** 2).sum(1)).mean())
        err = np.abs(b.astype(np.float64)[:, :3] - b64[:, :3]).max(1) / np.maximum(
            np.sqrt((b64[:, :3] ** 2).sum(1)), rscale)
        entry["f32_vs_f64_max_rel_pos_err"] = float(err.max())
        entry["r_scale"] = rscale
        e0, drift = energy_report(bodies, vel, G, dt, kmax)
        entry["E0"] = float(e0)
        entry["energy_drift"] = {k_: float(v_) for k_, v_ in drift.items()}
    manifest[name] = entry


def main():
    manifest = {"generator": "tests/golden/make_golden.py", "oracle": "oracle/nb_oracle.c",
                "layout": "row-major (n,4): bodies=x,y,z,m  vel=vx,vy,vz,0  accel=ax,ay,az,0"}
    # BASELINE.json config 1: N=1,024 Plummer, dt=1e-3, 100 steps, G=1
    b, v = ic.plummer(1024, seed=1)
    case(manifest, "plummer1024", b, v, 1e-3, 1.0, [1, 10, 100])
    # ragged N (not a multiple of the 256 tile; the reference is undefined there,
    # SURVEY.md §3.4) with unequal masses
    b, v = ic.uniform_cube(1000, seed=2)
    case(manifest, "cube1000", b, v, 1e-3, 1.0, [1, 20])
    # harsh mass ratio (1e7 : 10..50 as in generateGalaxy, nbody3d.js:62-64) and
    # the reference's default G = dt = 1e-4 (nbody3d.js:6-7), N=771
    rng = np.random.default_rng(3)
    n = 771
    b = np.zeros((n, 4), np.float32)
    v = np.zeros((n, 4), np.float32)
    b[0] = (0, 0, 0, 1e7)
    r = 0.5 + 2.5 * np.sqrt(rng.random(n - 1))
    th = rng.random(n - 1) * 2 * np.pi
    b[1:, 0] = r * np.cos(th)
    b[1:, 1] = r * np.sin(th)
    b[1:, 2] = (rng.random(n - 1) - 0.5) * 0.1
    b[1:, 3] = 10 + 40 * rng.random(n - 1)
    sp = np.sqrt(1e-4 * 1e7 / r)
    v[1:, 0] = -sp * np.sin(th)
    v[1:, 1] = sp * np.cos(th)
    case(manifest, "disk771", b, v, 1e-4, 1e-4, [1, 50])
    # reference-native initial conditions: produced by RUNNING the reference's
    # generateGalaxy text under node (make_galaxy_fixture.js; needs /root/reference),
    # stepped here with the reference's default G = dt = 1e-4 (nbody3d.js:6-7).
    # N = 789: not a multiple of 256, mass ratio 1e6 -- the harshest ordering/tail case.
    if os.path.isdir("/root/reference") and shutil.which("node"):
        subprocess.check_call(["node", os.path.join(HERE, "make_galaxy_fixture.js")])
    gb = np.fromfile(os.path.join(HERE, "galaxy_ref_bodies0.f32"), "<f4").reshape(-1, 4)
    gv = np.fromfile(os.path.join(HERE, "galaxy_ref_vel0.f32"), "<f4").reshape(-1, 4)
    case(manifest, "galaxy_ref", gb, gv, 1e-4, 1e-4, [1, 30])
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
