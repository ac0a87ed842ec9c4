"""No-GPU checks added in round 3: the reference-default initial conditions at full size, and what the release
build of the engine library must NOT contain."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, PKG
from nbody3d_amd import ic

CSRC = os.path.join(PKG, "csrc")


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_reference_default_system_matches_the_reference_generators_digest():
    """js/ic.js::galaxies at the reference's default UI state (index.html:68-74: 2 galaxies x 20,000 bodies -> N = 40,002,
    G = 1e-4) against the SHA-256 recorded when tests/golden/make_galaxy_fixture.js ran the reference's own generator
    text (nbody3d.js:51-133) on the same seeded stream: the full-size state every `-m gpu` galaxy test and bench.py's
    `also` entry start from is the reference generator's, bit for bit."""
    b, v, gp = ic.reference_galaxies(os.path.join(GOLDEN, "galaxy40002_params.json"))      # raises on a digest mismatch
    assert b.shape == (40002, 4) and v.shape == (40002, 4) and b.dtype == np.float32
    assert gp["n"] == 40002 and gp["minBodies"] == gp["maxBodies"] == 20000 and gp["G"] == 1e-4
    assert b[0, 3] == 1e7 and b[20001, 3] == 1e7                                           # nbody3d.js:62, one per galaxy
    assert np.all((b[1:20001, 3] >= 10) & (b[1:20001, 3] < 50)) and not v[:, 3].any()      # :63-64, :68,123
    assert np.allclose(b[:2].ravel(), gp["first_rows"]["bodies"]) and np.allclose(v[-2:].ravel(), gp["last_rows"]["vel"])
    sb, sv, sp = ic.reference_galaxies(os.path.join(GOLDEN, "galaxy_ref_params.json"))     # the committed small fixture, same route
    assert sb.tobytes() == open(os.path.join(GOLDEN, "galaxy_ref_bodies0.f32"), "rb").read()
    assert sv.tobytes() == open(os.path.join(GOLDEN, "galaxy_ref_vel0.f32"), "rb").read()


def test_release_library_reads_no_model_knobs_from_the_environment():
    """choose_shape called getenv("NB_MODEL_*") on every nb_create.  The constants are compiled in now;
    only the -DNB_TUNING calibration build (make tuning; tools/fit_model.py, fault injection) knows those names."""
    rel = open(os.path.join(CSRC, "libnbody3d_hip.so"), "rb").read()
    assert b"NB_MODEL_" not in rel and b"NB_TEST_FAIL" not in rel
    tun = os.path.join(CSRC, "libnbody3d_hip_tuning.so")
    if not os.path.exists(tun):
        subprocess.check_call(["make", "-C", CSRC, "-s", "tuning"])
    blob = open(tun, "rb").read()
    assert b"NB_MODEL_BOUNDARY" in blob and b"NB_TEST_FAIL_FRAME_SLOT" in blob


def _sym_plan(nsb, cps):
    H = (nsb - 1) // 2
    n_hi = 0 if nsb & 1 else nsb // 2
    total_hi, total_lo = (H + 1 + (1 if n_hi else 0)) * cps, (H + 1) * cps
    off = lambda g: g * total_hi if g <= n_hi else n_hi * total_hi + (g - n_hi) * total_lo      # noqa: E731
    return H, n_hi, total_hi, total_lo, off


@pytest.mark.parametrize("nsb,ranks,waves", [(2, 1, 4), (3, 1, 5), (8, 2, 7), (9, 3, 16), (16, 4, 12), (40, 8, 64), (41, 1, 100), (64, 8, 33)])
def test_symmetric_pass_partition_covers_every_pair_of_super_blocks_exactly_once(nsb, ranks, waves):
    """The index arithmetic of nb_force_symw / plan_launch (csrc/nb_kernels.hip.h, nb_plan.cpp), restated (tests/test_planner_cpu.py walks the planner's own output): super-blocks on a
    ring; super-block g sweeps the chunks of the H = (nsb-1)/2 super-blocks after it (and of the antipodal one when nsb is even
    and g < nsb/2), then its own in resident-only mode.  Rank r owns the super-blocks [r*nsb/ranks, (r+1)*nsb/ranks) and its
    waves cut THEIR lists, laid end to end, into floor/ceil-equal ranges.  Every unordered pair of different super-blocks must be
    swept by exactly one (rank, wave), every super-block's own block exactly once, every chunk exactly once."""
    cps = 4
    H, n_hi, total_hi, total_lo, off = _sym_plan(nsb, cps)
    if nsb % ranks:
        pytest.skip("ranks own whole super-blocks")
    seen_pairs, seen_diag, seen_chunks = {}, {}, set()
    for r in range(ranks):
        g0, g1 = r * nsb // ranks, (r + 1) * nsb // ranks
        p0, L = off(g0), off(g1) - off(g0)
        W = min(waves, L)
        for w in range(W):
            p, pend = p0 + w * L // W, p0 + (w + 1) * L // W
            while p < pend:
                first_lo = n_hi * total_hi
                if p < first_lo:
                    g, total = p // total_hi, total_hi
                    k = p - g * total_hi
                else:
                    g = n_hi + (p - first_lo) // total_lo
                    total = total_lo
                    k = (p - first_lo) - (g - n_hi) * total_lo
                assert g0 <= g < g1                                  # a rank never touches another rank's lists
                ring = total - cps
                kend = min(k + (pend - p), total)
                for kk in range(k, kend):
                    if kk < ring:
                        d = kk // cps
                        tb = (g + 1 + d) % nsb
                        assert d <= H and tb != g
                        seen_pairs[(frozenset((g, tb)), kk % cps)] = seen_pairs.get((frozenset((g, tb)), kk % cps), 0) + 1
                    else:
                        seen_diag[(g, kk - ring)] = seen_diag.get((g, kk - ring), 0) + 1
                    assert (g, kk) not in seen_chunks
                    seen_chunks.add((g, kk))
                p += kend - k
    # every unordered pair {a, b}, a != b: all cps traveler chunks of one of the two, swept by the OTHER one, exactly once
    for a in range(nsb):
        for b in range(a + 1, nsb):
            for c in range(cps):
                assert seen_pairs.get((frozenset((a, b)), c), 0) == 1, (a, b, c)
    assert len(seen_pairs) == nsb * (nsb - 1) // 2 * cps
    assert all(v == 1 for v in seen_diag.values()) and len(seen_diag) == nsb * cps
    assert len(seen_chunks) == n_hi * total_hi + (nsb - n_hi) * total_lo
