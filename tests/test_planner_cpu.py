"""The engine's launch planner on the host (nb_plan_query: no GPU needed).  These tests walk the plan the C++ planner built
the way the kernels walk it (nb_force_symw / nb_force_sym, nb_integrate_symw, csrc/nb_kernels.hip.h) and check the
properties the symmetric pass rests on: every unordered pair of bodies is evaluated exactly once, no partial-sum row is
written twice, and the integrate kernel reads exactly the rows that were written.  The reference has no counterpart (its
dispatch is ceil(N / 256) workgroups, nbody3d.js:478); the path these plans serve is nbody3d.js:245-272."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from nbody3d_amd import capi

NB_FLAG_NO_SYM, NB_FLAG_SYM_SHARD, NB_FLAG_WHOLE_SWEEPS = 64, 128, 256


def walk_symw(q, n):
    """Every chunk-sweep of a whole-system wave-granular plan as arrays (one entry per list position), and the waves that share it:
    the plan cuts the L sweeps into W ranges of UNITS (ups units = 64 rotation steps per sweep).

    The ring holds the nsb WHOLE super-blocks; a ragged n leaves a short block Z of zc real chunks behind them, which every
    super-block sweeps after its ring sweeps ("z sweeps", both sides) and which sweeps only its own chunks, last in the list."""
    pl, tab = q["plan"], q["tab"]
    J = 2 if q["x"] == 1 else 1
    S, CH = 64 * q["ipl"], 64 * J
    cps = S // CH
    nsb, W, L, zc, ups = pl["nsb"], pl["W"], pl["L"], pl["zc"], pl["ups"]
    assert ups in (1, 2, 4, 8, 16, 32, 64) and q["ups"] == ups
    blocks = -(-n // S)
    assert pl["np"] == blocks * S and nsb == n // S and zc == -(-(n % S) // CH) and blocks == nsb + (1 if zc else 0)
    H, n_hi = pl["H"], pl["n_hi"]
    assert H == (nsb - 1) // 2 and n_hi == (0 if nsb % 2 else nsb // 2)
    assert pl["total_lo"] == (H + 1) * cps + zc and pl["total_hi"] == (H + 1 + (1 if n_hi else 0)) * cps + zc
    first_lo = n_hi * pl["total_hi"]
    first_z = first_lo + (nsb - n_hi) * pl["total_lo"]
    assert L == first_z + zc
    p = np.arange(L, dtype=np.int64)
    hi, zown = p < first_lo, p >= first_z
    g = np.where(hi, p // pl["total_hi"], np.where(zown, nsb, n_hi + (p - first_lo) // pl["total_lo"]))
    k = np.where(hi, p - g * pl["total_hi"], np.where(zown, p - first_z, (p - first_lo) - (g - n_hi) * pl["total_lo"]))
    total = np.where(hi, pl["total_hi"], np.where(zown, zc, pl["total_lo"]))
    ring = np.where(zown, 0, total - cps - zc)
    both_end = np.where(zown, 0, ring + zc)
    ringsw = k < ring                              # over a chunk of another whole super-block
    zsw = (k >= ring) & (k < both_end)             # over a chunk of Z
    sym = ringsw | zsw                             # both keep traveler sums
    d = k // cps
    tb = np.where(ringsw, (g + 1 + d) % max(nsb, 1), np.where(zsw, nsb, g))
    c = np.where(ringsw, k % cps, np.where(zsw, k - ring, k - both_end))
    tstart = tb * S + c * CH
    assert np.all(tstart < n)                      # no sweep over padding only
    # the waves: W ranges ("positions", in list order) of the L * ups units, none of them empty, each run by ONE physical wave (a record of
    # the plan's table: {first unit, end, resident layer, spill row}).  One wave per SIMD: position = wave, the ranges equal in WORK to about
    # one unit -- a sweep over an own chunk (no traveler sums) counts 7, any other 8.  Two waves per SIMD: positions 2i / 2i + 1 belong to
    # waves i and W / 2 + i -- the OLDER and the younger wave of a SIMD -- and the older one has 0.92 (f64: 0.85) of the pair's work
    Lu = L * ups
    starts = q["starts"].astype(np.int64)
    order = q["order"].astype(np.int64)
    assert sorted(order.tolist()) == list(range(W))
    # a record ends where the wave's OWN part ends: at the next range's start, or (whole sweeps, two waves per SIMD) a few sweeps short of
    # it -- those sweeps are in the queue that every wave draws from when its own part is done, each exactly once
    own_end = q["waves"][order, 1].astype(np.int64)
    pieces = q["pieces"].astype(np.int64)
    tails = starts[1:] - own_end
    assert np.all(tails >= 0) and np.all(own_end > starts[:-1]) and tails.sum() == (pieces[:, 1] >> 16).sum()
    if len(pieces):
        assert ups == 1 and np.all(tails[1::2] == 0)
        # a piece = (first unit, resident layer | sweeps << 16): whole sweeps inside ONE super-block's list; together the tails, each sweep once
        plen = pieces[:, 1] >> 16
        covered = np.concatenate([np.arange(a0, a0 + l0) for a0, l0 in zip(pieces[:, 0], plen)])
        want_pieces = np.concatenate([np.arange(own_end[pp], starts[pp + 1]) for pp in range(W)])
        assert sorted(covered.tolist()) == want_pieces.tolist() and plen.min() >= 1
        assert np.all(g[pieces[:, 0]] == g[pieces[:, 0] + plen - 1])
        assert plen[0] >= plen[-1] and plen[-1] == 1                                                 # long pieces first, single sweeps last
        assert 0.02 < tails[0::2].sum() / (starts[1::2] - starts[0:-1:2]).sum() < 0.05          # ~3.5 % of the older waves' ranges
    else:
        assert np.all(tails == 0)
    assert len(starts) == W + 1 and starts[0] == 0 and starts[-1] == Lu and W <= Lu and np.diff(starts).min() >= 1
    work = np.repeat(np.where(sym, 8, 7 if ups > 1 else 8).astype(np.int64), ups)              # per unit (whole sweeps: an even cut)
    per_pos = np.add.reduceat(work, starts[:-1])
    assert per_pos.sum() == work.sum()
    paired = not np.array_equal(order, np.arange(W))
    if paired:
        assert W % 8 == 0 and np.array_equal(order[0::2], np.arange(W // 2)) and np.array_equal(order[1::2], W // 2 + np.arange(W // 2))
        old, young = per_pos[0::2], per_pos[1::2]
        assert old.max() - old.min() <= 24 and young.max() - young.min() <= 24, (old.min(), old.max(), young.min(), young.max())
        share = old.sum() / per_pos.sum()
        assert abs(share - (0.85 if q["variant"].startswith("f64") else 0.92)) < 0.02, share
        pair = old + young
        assert pair.max() - pair.min() <= 16
    else:
        assert per_pos.max() - per_pos.min() <= 16, (per_pos.min(), per_pos.max())
    pos_u = np.searchsorted(starts, np.arange(Lu, dtype=np.int64), side="right") - 1            # the position of every unit
    assert np.all((pos_u >= 0) & (pos_u < W))
    wu = order[pos_u]                            # ... and its wave (a queued sweep: the wave whose range it was cut from; any wave may run it)
    queued = np.zeros(Lu, bool)
    for a0, l0 in zip(pieces[:, 0], pieces[:, 1] >> 16):
        queued[a0:a0 + l0] = True
    w = wu[::ups]                                # the wave that starts each sweep (steps from 0): it owns the sweep's traveler layer / z-row
    w_last = wu[ups - 1::ups]                    # ... and the one that ends it
    return dict(S=S, CH=CH, cps=cps, nsb=nsb, blocks=blocks, zc=zc, H=H, n_hi=n_hi, g=g, k=k, sym=sym, ringsw=ringsw, zsw=zsw, d=d, tb=tb, c=c,
                tstart=tstart, w=w, w_last=w_last, wu=wu, pos_u=pos_u, order=order, starts=starts, ups=ups, pl=pl, tab=tab, queued=queued, pieces=pieces)


def check_spill_lists(q, n, wk):
    """ups > 1: a wave whose range starts INSIDE a sweep keeps that sweep's traveler sums in its own spill row; the plan lists,
    per traveler chunk, the waves K2 has to add -- exactly those, in ascending order."""
    ups, W, CH, starts = wk["ups"], wk["pl"]["W"], wk["CH"], wk["starts"]
    zrows = wk["nsb"] * wk["zc"]                  # the z-rows come first in the spill buffer: whole super-block g's sums for chunk c of Z = row g * zc + c
    if ups == 1:
        assert q["spill_rows"] == zrows * CH and "spill_tab" not in q
        return
    st, ids, slot = q["spill_tab"], q["spill_ids"], q["spill_slot"]
    assert q["spill_rows"] == (zrows + max(1, len(ids))) * CH and len(slot) == W
    assert all(int(slot[int(wv)]) == zrows + e for e, wv in enumerate(ids))        # spill row zrows + e belongs to wave ids[e]: rows run chunk by chunk
    assert st.shape[0] == wk["pl"]["np"] // CH
    want = {}
    for p in range(W):                            # in list order
        u0 = int(starts[p])
        if u0 % ups == 0:
            continue
        sw = u0 // ups
        if wk["sym"][sw]:
            want.setdefault(int(wk["tstart"][sw]) // CH, []).append(int(wk["order"][p]))
    got = {}
    for ci in range(st.shape[0]):
        off, cnt = int(st[ci, 0]) - zrows, int(st[ci, 1])
        if cnt:
            got[ci] = [int(x) for x in ids[off:off + cnt]]
    assert got == want
    assert sum(len(v) for v in want.values()) == len(ids)
    # a sweep is shared by consecutive positions: the first owns the layer, each of the others spills exactly once (rows in list order)


def check_whole_plan(q, n):
    """A whole-system wave-granular plan: pair coverage, layer writes, and K2's read set."""
    wk = walk_symw(q, n)
    nsb, blocks, zc, cps, H, n_hi, pl, tab = wk["nsb"], wk["blocks"], wk["zc"], wk["cps"], wk["H"], wk["n_hi"], wk["pl"], wk["tab"]
    g, sym, ringsw, zsw, d, tb, c, w = wk["g"], wk["sym"], wk["ringsw"], wk["zsw"], wk["d"], wk["tb"], wk["c"], wk["w"]
    assert tab.shape[0] == blocks
    # (1) unordered pairs of DIFFERENT whole super-blocks: each (resident block, traveler chunk) at most once, and for a != b
    #     either a sweeps all of b's chunks or b all of a's -- never both, never neither
    visits = np.zeros((nsb, nsb, cps), np.int32)
    np.add.at(visits, (g[ringsw], tb[ringsw], c[ringsw]), 1)
    assert visits.max() <= 1
    per_pair = visits.sum(axis=2)
    assert np.all(np.diag(per_pair) == 0)
    both = per_pair + per_pair.T
    off = ~np.eye(nsb, dtype=bool)
    assert np.all(both[off] == cps) and np.all((per_pair[off] == 0) | (per_pair[off] == cps))
    # (1z) the short block Z: EVERY whole super-block sweeps each of its zc real chunks exactly once, from both sides
    zv = np.zeros((nsb, max(zc, 1)), np.int32)
    np.add.at(zv, (g[zsw], c[zsw]), 1)
    assert zsw.sum() == nsb * zc and (zc == 0 or np.all(zv == 1)) and np.all(tb[zsw] == nsb)
    # (2) pairs INSIDE a block: its own chunks once each (Z: its zc real ones), resident-only
    own = np.zeros((blocks, cps), np.int32)
    np.add.at(own, (g[~sym], c[~sym]), 1)
    assert np.all(own[:nsb] == 1) and np.all(tb[~sym] == g[~sym])
    if zc:
        assert np.all(own[nsb, :zc] == 1) and np.all(own[nsb, zc:] == 0)
    # (3) ring distances stay inside the traveler layers
    assert np.all(d[ringsw] < H + (g[ringsw] < n_hi)) and pl["t_layer0"] + H + (1 if n_hi else 0) == q["sym_layers"]
    # (4) resident layers.  The waves whose range ENDS in b's list add their sums up per workgroup of four (LDS) and write the layer
    #     their table records name; the last position of the list, if its range goes on into b + 1, writes its part to b's last
    #     layer.  The table = {first wave, layer count}: every layer K2 reads is written exactly once
    gu = np.repeat(g, wk["ups"])
    starts, order, waves, queued = wk["starts"], wk["order"], q["waves"], wk["queued"]
    piece_layer = {int(u): int(l) & 0xffff for u, l in wk["pieces"]}
    for b in range(blocks):
        units = np.nonzero(gu == b)[0]
        own = units[~queued[units]]              # the units of b's list that belong to a wave's own part
        written = []
        goes_on = False
        if len(own):
            ps = np.unique(wk["pos_u"][own])
            assert order[np.unique(wk["pos_u"][units])[0]] == tab[b, 0], b
            last_end = int(waves[order[ps[-1]], 1])
            goes_on = last_end > units[-1] + 1       # the last own part runs on into b + 1
            ending = ps[:-1] if goes_on else ps
            layer_of_wg = {}
            for pp in ending:
                wv = int(order[pp])
                assert layer_of_wg.setdefault(wv // 4, int(waves[wv, 2])) == int(waves[wv, 2]), (b, wv)      # one row set per workgroup
            written = sorted(layer_of_wg.values())
        written += [piece_layer[int(u)] for u in units[queued[units]] if int(u) in piece_layer]      # every queued piece of b's list: a layer of its own
        if goes_on:
            written.append(int(tab[b, 1]) - 1)
        assert sorted(written) == list(range(int(tab[b, 1]))), (b, written, tab[b])
    assert pl["r_layer0"] == 0 and tab[:, 1].max() == pl["t_layer0"] == q["jsplit"]
    # (5) traveler layers: K2 (nb_integrate_symw) reads layers t_layer0 + [0, H + (n_hi and b >= n_hi)) of every row of a whole
    #     super-block b -- exactly the set written, once each; for the rows of Z it reads the z-rows g * zc + c of all g instead:
    #     written once each by (1z)
    tl = np.zeros((nsb, cps, H + 1), np.int32)
    np.add.at(tl, (tb[ringsw], c[ringsw], d[ringsw]), 1)
    for b in range(nsb):
        nt = H + (1 if (n_hi and b >= n_hi) else 0)
        assert np.all(tl[b, :, :nt] == 1) and np.all(tl[b, :, nt:] == 0), b
    # (6) sweeps shared by several waves: the later parts go through the spill lists
    check_spill_lists(q, n, wk)
    return wk


SIZES = [8193, 13000, 14000, 16384, 20000, 24001, 32768, 40002, 65536, 100000, 131072, 262144]


@pytest.mark.parametrize("n", SIZES)
def test_default_plan_covers_every_pair_once(n):
    q = capi.plan_query(n)
    if not q["sym"]:
        pytest.skip("planner keeps an ordered-pair kernel at N=%d: %s" % (n, q["variant"]))
    assert q["symw"] and q["variant"].startswith("f32pk_symw_")
    check_whole_plan(q, n)


@pytest.mark.parametrize("variant", [704013, 708013, 708011, 716013, 716011])
@pytest.mark.parametrize("n,jsplit", [(2049, 0), (9001, 1), (30000, 2), (33000, 3), (70001, 0)])
def test_pinned_wave_granular_plans(variant, n, jsplit):
    q = capi.plan_query(n, force_variant=variant, jsplit=jsplit)
    S = 64 * (variant // 1000 % 100)
    if n <= S:
        assert not q["sym"]
        return
    assert q["symw"] and q["ipl"] == variant // 1000 % 100 and q["x"] == variant % 10
    wk = check_whole_plan(q, n)
    assert wk["pl"]["W"] <= wk["pl"]["L"] * wk["pl"]["ups"]              # never more waves than units: no wave without work
    whole = capi.plan_query(n, force_variant=variant, jsplit=jsplit, flags=NB_FLAG_WHOLE_SWEEPS)       # the A/B arm: whole sweeps per wave
    assert whole["ups"] == 1 and "_u" not in whole["variant"].rsplit("_r", 1)[1]
    check_whole_plan(whole, n)


@pytest.mark.parametrize("n", [40002, 262144])
def test_f64_plan(n):
    q = capi.plan_query(n, precision="f64")
    assert q["variant"].startswith("f64_symw_ipl8_j1")
    check_whole_plan(q, n)


@pytest.mark.parametrize("n,jsplit", [(4097, 0), (20000, 3), (40002, 0), (65536, 16), (262144, 0)])
def test_workgroup_form_plan(n, jsplit):
    """nb_force_sym<4,4,2>: workgroup (g, q) sweeps chunks [q total / Q, (q+1) total / Q) of g's list."""
    q = capi.plan_query(n, force_variant=708014, jsplit=jsplit)
    assert q["sym"] and not q["symw"] and q["variant"].startswith("f32pk_sym_ipl8_ws4")
    pl = q["plan"]
    S, CH = 2048, 128
    cps, nsb, Q, H, n_hi = S // CH, pl["nsb"], pl["q"], pl["H"], pl["n_hi"]
    assert nsb == -(-n // S) and pl["np"] == nsb * S and 1 <= Q <= pl["total_hi"]
    visits = np.zeros((nsb, nsb, cps), np.int32)
    own = np.zeros((nsb, cps), np.int32)
    for g in range(nsb):
        ring = (H + (1 if g < n_hi else 0)) * cps
        total = ring + cps
        assert total == (pl["total_hi"] if g < n_hi else pl["total_lo"])
        cuts = [(s * total) // Q for s in range(Q + 1)]
        assert cuts[0] == 0 and cuts[-1] == total and all(b >= a for a, b in zip(cuts, cuts[1:]))
        for k in range(total):
            if k < ring:
                visits[g, (g + 1 + k // cps) % nsb, k % cps] += 1
            else:
                own[g, k - ring] += 1
    per_pair = visits.sum(axis=2)
    off = ~np.eye(nsb, dtype=bool)
    assert visits.max() <= 1 and np.all(own == 1) and np.all((per_pair + per_pair.T)[off] == cps)
    assert pl["t_layer0"] == Q and q["sym_layers"] == Q + H + (1 if n_hi else 0)


def walk_rank(q, n):
    """The two phases of a rank-form plan as the kernel walks them: per phase (g, k, wave that starts the sweep, wave of every unit, first unit of every wave)."""
    rp, S = q["rank_plan"], 64 * q["ipl"]
    cps, ups = S // 64, rp["ups"]
    g0, g1, ng = rp["g0"], rp["g1"], rp["g1"] - rp["g0"]
    out = {}
    for phase, L, W, pre in (("A", rp["LA"], rp["WA"], q["prefix_a"]), ("B", rp["LB"], rp["WB"], q["prefix_b"])):
        assert len(pre) == ng + 1 and pre[0] == 0 and pre[-1] == L and np.all(np.diff(pre.astype(np.int64)) >= 0)
        if L == 0:
            assert W == 0
            out[phase] = (np.zeros(0, np.int64),) * 5
            continue
        assert 1 <= W <= L                                    # at least one sweep's worth of units per wave
        p = np.arange(L, dtype=np.int64)
        gi = np.searchsorted(pre.astype(np.int64), p, side="right") - 1
        g = g0 + gi
        j = p - pre[gi]
        total = np.where(g < rp["n_hi"], rp["total_hi"], rp["total_lo"])
        ring = total - cps
        a = np.minimum(ring, (g1 - 1 - g) * cps)
        k = (a + j) if phase == "B" else np.where(j < a, j, ring + (j - a))
        assert np.all(k < total)
        Lu = L * ups
        starts = (np.arange(W + 1, dtype=np.int64) * Lu) // W
        assert np.diff(starts).min() >= 1 and np.diff(starts).max() - np.diff(starts).min() <= 1      # balanced to one UNIT
        wu = np.searchsorted(starts, np.arange(Lu, dtype=np.int64), side="right") - 1
        out[phase] = (g, k, wu[::ups], wu, starts)
    return out


@pytest.mark.parametrize("n,g,prec", [(8192, 2, "f32"), (65536, 8, "f32"), (262144, 8, "f32"), (262144, 3, "f32"), (1048576, 8, "f32"), (40960, 5, "f32"),
                                     (16384, 4, "f64"), (262144, 8, "f64")])
def test_rank_form_plans_tile_the_pair_list(n, g, prec):
    """NB_FLAG_SYM_SHARD: rank r sweeps the chunk lists of its own super-blocks only -- the ranks together evaluate every unordered
    pair exactly once -- in two phases: A = the sweeps whose travelers are the rank's OWN rows (what an overlapped step issues before
    it waits for the all-gather), B = the rest."""
    align = 512 if prec == "f64" else 1024
    rows = -(-(-(-n // g)) // align) * align
    whole = capi.plan_query(n, precision=prec, force_variant=(708013 if prec == "f64" else 716013), jsplit=1, flags=NB_FLAG_WHOLE_SWEEPS)
    assert whole["symw"]
    wpl = whole["plan"]
    seen = np.zeros(wpl["L"], np.int32)                      # every position of the whole system's list, by (g, k)
    off = lambda gg: np.where(gg <= wpl["n_hi"], gg * wpl["total_hi"], wpl["n_hi"] * wpl["total_hi"] + (gg - wpl["n_hi"]) * wpl["total_lo"])
    for r in range(g):
        b = min(r * rows, n)
        cnt = min(rows, n - b)
        if cnt == 0:
            continue
        q = capi.plan_query(n, precision=prec, shard=(b, cnt), flags=NB_FLAG_SYM_SHARD)
        assert q["sym_rank"] and "symwrank" in q["variant"], q["variant"]
        rp, pl = q["rank_plan"], q["plan"]
        assert {k: rp[k] for k in ("np", "nsb", "total_hi", "total_lo", "n_hi", "H")} == {k: wpl[k] for k in ("np", "nsb", "total_hi", "total_lo", "n_hi", "H")}
        S = 64 * q["ipl"]
        cps = S // 64
        assert rp["g0"] == q["sym_g0"] == b // S and rp["g1"] == q["sym_g1"] == (b + cnt) // S
        ups = rp["ups"]
        assert pl["L"] == rp["LA"] + rp["LB"] and pl["W"] == rp["WA"] + rp["WB"] and pl["ups"] == ups and q["ups"] == ups
        assert q["spill_rows"] == (0 if ups == 1 else 64 * (rp["WA"] + rp["WB"]))
        assert q["own_splits"] == rp["WA"] and q["own_split0"] == 0          # what nb_shape_info reports: the waves issued before the wait
        wk = walk_rank(q, n)
        want_spill = {}
        for phase in "AB":
            gg, kk, ww, wu, starts = wk[phase]
            if len(gg) == 0:
                continue
            np.add.at(seen, off(gg) + kk, 1)
            total = np.where(gg < rp["n_hi"], rp["total_hi"], rp["total_lo"])
            sym = kk < total - cps
            tb = np.where(sym, (gg + 1 + kk // cps) % rp["nsb"], gg)
            own = (tb >= rp["g0"]) & (tb < rp["g1"])
            if phase == "A":
                assert np.all(own)                            # residents AND travelers are the rank's own rows: nothing of the gather is read
            elif g > 1:
                assert not np.any(own & (tb > gg))            # an own target ahead on the ring would have been phase A's
            # the table: first wave and wave count of every own super-block in this phase
            col = 0 if phase == "A" else 2
            # a wave that starts inside a symmetric sweep over real rows spills it: B waves numbered from WA
            for wv in range(len(starts) - 1):
                u0 = int(starts[wv])
                if u0 % ups:
                    sw = u0 // ups
                    if sym[sw] and int(tb[sw]) * S + int(kk[sw] % cps) * 64 < n:
                        want_spill.setdefault(int(tb[sw]) * (S // 64) + int(kk[sw] % cps), []).append(wv + (rp["WA"] if phase == "B" else 0))
            gu = np.repeat(gg, ups)
            for sb in range(rp["g0"], rp["g1"]):
                ws = np.unique(wu[gu == sb])
                if len(ws):
                    assert ws[0] == q["rank_tab"][sb, col] and len(ws) == q["rank_tab"][sb, col + 1] and ws[-1] - ws[0] + 1 == len(ws)
                else:
                    assert q["rank_tab"][sb, col + 1] == 0
        assert rp["r_layer0"] == 0 and rp["rb_layer0"] == q["rank_tab"][:, 1].max() and rp["t_layer0"] == rp["rb_layer0"] + q["rank_tab"][:, 3].max()
        assert q["sym_layers"] == rp["t_layer0"] + rp["H"] + (1 if rp["n_hi"] else 0)
        if ups > 1:
            st, ids = q["spill_tab"], q["spill_ids"]
            got = {ci: [int(x) for x in ids[int(st[ci, 0]):int(st[ci, 0]) + int(st[ci, 1])]] for ci in range(st.shape[0]) if st[ci, 1]}
            assert got == {k2: sorted(v2) for k2, v2 in want_spill.items()} and sum(len(v2) for v2 in got.values()) == len(ids)
        else:
            assert not want_spill
        if g > 1:
            assert 0.5 / g < rp["LA"] / pl["L"] < 2.0 / g       # about 1 / ranks of the work runs before the wait
    assert np.all(seen == 1)                                     # the ranks' phases tile the system's pair list


@pytest.mark.parametrize("n,prec,budget", [(100000, "f32", 48), (131072, "f32", 64), (65536, "f64", 48), (2500000, "f32", 16384), (4194304, "f32", 0)])
def test_layer_budget_passes_tile_the_pair_list(n, prec, budget):
    """A whole system whose traveler layers would not fit the layer budget: the rank-form pipeline on one device ("local": no
    communicator), the ring distances in passes that reuse the layers.  Walked pass by pass as the kernel does: every position of
    every super-block's list exactly once, the traveler layers a pass writes lie inside its window, and the layers fit the budget."""
    q0 = capi.plan_query(n, precision=prec, layer_budget_mib=budget)
    assert q0["sym_rank"] and q0["local"] and q0["passes"] >= 2 and q0["variant"].endswith("_p%d" % q0["passes"]), q0["variant"]
    esz = 8 if prec == "f64" else 4
    limit = budget * 2**20 if budget else min(288e9 / 3, 96 * 2**30)
    assert 3 * esz * q0["sym_np"] * q0["sym_layers"] <= limit
    S = 64 * q0["ipl"]
    cps = S // 64
    rp0 = q0["rank_plan"]
    nsb, n_hi, H = rp0["nsb"], rp0["n_hi"], rp0["H"]
    totals = np.where(np.arange(nsb) < n_hi, rp0["total_hi"], rp0["total_lo"])
    seen = [np.zeros(t, np.int32) for t in totals]
    if nsb > 512:                 # big systems: check a sample of super-blocks (the walk is O(sweeps))
        sample = set(np.linspace(0, nsb - 1, 64).astype(int).tolist())
    else:
        sample = set(range(nsb))
    for ps in range(q0["passes"]):
        q = capi.plan_query(n, precision=prec, layer_budget_mib=budget, sym_pass=ps)
        rp = q["rank_plan"]
        assert rp["g0"] == 0 and rp["g1"] == nsb and rp["ups"] == 1 and q["spill_rows"] == 0
        k_lo, k_hi, d0 = q["pass_k_lo"], q["pass_k_hi"], q["pass_d0"]
        assert k_lo == d0 * cps and (ps == 0) == (k_lo == 0) and (ps == q0["passes"] - 1) == (k_hi == 0xffffffff)
        d1 = (k_hi // cps) if k_hi != 0xffffffff else H + (1 if n_hi else 0)
        assert rp["t_layer0"] + (d1 - d0) <= q0["sym_layers"]
        for phase, L, W, pre in (("A", rp["LA"], rp["WA"], q["prefix_a"]), ("B", rp["LB"], rp["WB"], q["prefix_b"])):
            assert len(pre) == nsb + 1 and pre[0] == 0 and pre[-1] == L and (W == min(4 * 256 * 2, L))
            for g in sample:
                ring = totals[g] - cps
                a = min(ring, (nsb - 1 - g) * cps)
                a_lo, a_len = min(a, k_lo), min(a, k_hi) - min(a, k_lo)
                b_lo = min(ring, max(a, k_lo))
                ln = int(pre[g + 1]) - int(pre[g])
                j = np.arange(ln)
                k = (b_lo + j) if phase == "B" else np.where(j < a_len, a_lo + j, ring + (j - a_len))
                assert np.all(k < totals[g])
                sym = k < ring
                assert np.all((k[sym] >= k_lo) & (k[sym] < k_hi))                  # ring sweeps inside the window: layers t_layer0 + d - d0
                seen[g][k] += 1
    for g in sample:
        assert np.all(seen[g] == 1), g


def test_rank_form_needs_whole_super_blocks():
    assert not capi.plan_query(262144, shard=(1000, 32768), flags=NB_FLAG_SYM_SHARD)["sym"]           # unaligned rows
    assert not capi.plan_query(262144, shard=(0, 32768), flags=NB_FLAG_SYM_SHARD | NB_FLAG_NO_SYM)["sym"]
    assert not capi.plan_query(262144, shard=(0, 32768))["sym"]                                      # a plain shard: ordered pairs
    assert capi.plan_query(262144, precision="f64", shard=(0, 32768), flags=NB_FLAG_SYM_SHARD)["ipl"] == 8


@pytest.mark.parametrize("n", [1, 2, 255, 1024, 4000, 8192, 12000, 40002, 262144])
@pytest.mark.parametrize("flags", [NB_FLAG_NO_SYM, NB_FLAG_NO_SYM | 4, NB_FLAG_NO_SYM | 8])
def test_ordered_pair_plans_partition_j(n, flags):
    """The ordered-pair forms: the j-partitions cover [0, n) (multiples of 8 bodies, at most 128 of them)."""
    q = capi.plan_query(n, flags=flags)
    assert not q["sym"] and 1 <= q["jsplit"] <= 128
    if q["kind"] == 6:          # j-packed fused step: whole 4-pair units per wave
        assert q["jsplit"] * q["j_per_split"] >= n
    else:
        assert q["j_per_split"] % 8 == 0 and (q["jsplit"] - 1) * q["j_per_split"] < n <= q["jsplit"] * q["j_per_split"]
    assert q["own_splits"] == 0


@pytest.mark.parametrize("n,g", [(262144, 2), (262144, 4), (262144, 8), (1048576, 8), (100000, 3)])
def test_shard_own_splits_lie_inside_the_shard(n, g):
    """Plain i-shards (ordered pairs): the j-partitions the overlapped exchange issues first are whole and inside the shard."""
    rows = -(-n // g)
    for r in range(g):
        b, cnt = r * rows, min(rows, n - r * rows)
        q = capi.plan_query(n, shard=(b, cnt))
        lo, hi = q["own_split0"] * q["j_per_split"], (q["own_split0"] + q["own_splits"]) * q["j_per_split"]
        if q["own_splits"]:
            assert lo >= b and min(hi, n) <= b + cnt
            assert lo - b < q["j_per_split"] and (b + cnt) - min(hi, n) < q["j_per_split"]      # and as many as fit
        else:
            assert cnt < 2 * q["j_per_split"]


def test_model_choice_table():
    """The automatic choice at the sizes DESIGN.md quotes (256 CUs, 2.4 GHz): a change of the cost model shows up here."""
    want = {1024: "f32pk_fused_regs1024_ipl2_ls64", 6500: "f32pk_fused_lds2048_ipl2_ls32", 7000: "f32pk_symw_ipl8_j1_w1024_r22t6_u32", 8192: "f32pk_symw_ipl8_j1_w1024_r19t8_u32", 9000: "f32pk_symw_ipl8_j1_w1024_r16t8_u32", 10000: "f32pk_symw_ipl16_j1_w1024_r30t4_u32", 11000: "f32pk_symw_ipl16_j1_w1012_r27t5",
            13000: "f32pk_symw_ipl16_j1_w1024_r25t6_u32", 16384: "f32pk_symw_ipl16_j1_w1024_r19t8_u32", 20000: "f32pk_symw_ipl16_j1_w2048_r31t9_u32", 40002: "f32pk_symw_ipl16_j1_w2048_r17t19_u8", 65536: "f32pk_symw_ipl16_j1_w2048", 262144: "f32pk_symw_ipl16_j1_w2048_r32t128", 1048576: "f32pk_symw_ipl16_j1_w2048"}
    for n, prefix in want.items():
        assert capi.plan_query(n)["variant"].startswith(prefix), (n, capi.plan_query(n)["variant"])
    assert capi.plan_query(262144, precision="f64")["variant"].startswith("f64_symw_ipl8_j1_w2048")
    # the layers grow with N^2: past the budget (a third of the device memory, at most 96 GiB; nb_config.layer_budget_mib) the ring
    # distances go in passes that reuse the layers (the rank-form pipeline on one device, "_pN"); ordered pairs only when not even one
    # distance per pass fits
    assert capi.plan_query(2500000)["variant"].startswith("f32pk_symw_ipl16") and capi.plan_query(2500000, layer_budget_mib=16384)["variant"].endswith("_p3")
    assert capi.plan_query(4500000)["variant"].startswith("f32pk_symwrank_ipl16") and "_p" in capi.plan_query(4500000)["variant"]
    assert capi.plan_query(1048576, precision="f64")["sym"] == 1 and capi.plan_query(40002, layer_budget_mib=4)["sym"] == 0


def test_bad_arguments_are_errors():
    with pytest.raises(capi.NBodyError):
        capi.plan_query(0)
    with pytest.raises(capi.NBodyError):
        capi.plan_query(1000, shard=(900, 200))
    with pytest.raises(capi.NBodyError):
        capi.plan_query((1 << 30) + 1)
    assert capi.plan_query(1 << 30)["sym"] == 0            # the largest system the 32-bit row arithmetic takes: ordered pairs


def test_planner_under_address_and_ub_sanitizers(tmp_path):
    """nb_plan.cpp is host-only C++: compiled with -fsanitize=address,undefined and driven over ~170,000 configurations."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    csrc = os.path.join(root, "nbody3d-webgpu_amd", "csrc")
    exe = str(tmp_path / "plan_sanitize")
    cc = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", csrc,
                         os.path.join(root, "tests", "c", "plan_sanitize.cpp"), os.path.join(csrc, "nb_plan.cpp"), "-o", exe],
                        capture_output=True, text=True, timeout=300)
    if cc.returncode != 0 and ("asan" in cc.stderr or "ubsan" in cc.stderr or "sanitize" in cc.stderr):
        pytest.skip("this g++ has no sanitizer runtime: " + cc.stderr[-200:])
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0 and "under ASan + UBSan" in run.stdout, run.stdout[-1000:] + run.stderr[-3000:]


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
