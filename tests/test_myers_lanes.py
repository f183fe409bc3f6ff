"""One alignment per lane (csrc/hip/rtk_myers_lane.h, stage entry rtk_myers_batch_lanes): distance, end locations and path of edlibAlign for the small problems
of the region program, held to the reference's golden vectors (plain equality on those without IUPAC codes, IUPAC equality on all of them) and to the oracle on random problems --
plain, with other characters and with IUPAC codes in the query, thresholds k, the three modes, zero lengths, and the problems the route hands on to the wave route (query above 512
characters, a target character outside ACGTN). The code of a lane has no cross-lane operation, so the 1-lane simulator runs exactly what a lane of
the device runs; the gpu test runs the same problems 64 per wavefront."""
import random

import pytest

from conftest import SIM_LIB, golden_rows
from oracle import oracle_py as op
from ratatosk_amd import api


def _problems():
    rnd = random.Random(11)
    qs, ts, ks, ms = [], [], [], []

    def mutate(s, rate, alpha):
        out = []
        for c in s:
            r = rnd.random()
            if r < rate / 3:
                continue
            if r < 2 * rate / 3:
                out.append(rnd.choice(alpha))
            if r < rate:
                out.append(rnd.choice(alpha)); continue
            out.append(c)
        return "".join(out)
    for i in range(700):
        alpha = "ACGT" if i % 5 else ("ACGTN" if i % 10 else "ACGTNRYKMSW")  # the last one: target characters that are not for the lane route
        m = rnd.choice((1, 5, 31, 63, 64, 65, 100, 128, 129, 200, 255, 256, 300, 511, 512, 513, 700))
        q = "".join(rnd.choice(alpha) for _ in range(m))
        mode = i % 3
        if mode == 0:
            t = mutate(q, rnd.choice((0.0, 0.05, 0.15, 0.4)), alpha)
        else:
            core = mutate(q, rnd.choice((0.0, 0.1, 0.3)), alpha)
            t = "".join(rnd.choice(alpha) for _ in range(rnd.randrange(0, 80))) + core + "".join(rnd.choice(alpha) for _ in range(rnd.randrange(0, 80)))
        if i % 97 == 0:
            t = ""
        if i % 101 == 0:
            q = ""
        qs.append(q); ts.append(t); ms.append(mode)
        ks.append(-1 if i % 4 else rnd.choice((0, 3, 10, 40, 200)))
    return qs, ts, ks, ms


def _fits_lane_route(q, t, k, mode, want_path):
    """What rtk_myers_lane takes (csrc/hip/rtk_myers_lane.h): queries up to 512 characters, targets up to 2048 over A C G T N; with a path, a table of at most 4096
    word-columns for target[0 .. first end location] (bounded here by the whole target) and a move list that fits. Zero lengths are answered by the route itself."""
    if len(q) == 0 or len(t) == 0:
        return True
    if len(q) > 512 or len(t) > 2048:
        return False
    if mode == 0 and k >= 0 and k < abs(len(t) - len(q)):
        return True  # answered before the sweep (edlib.cpp:744-747)
    if not set(t) <= set("ACGTN"):
        return False
    return True if not want_path else None  # paths: the table bound depends on the end location, so only the lower bound below is held


def _assert_routes(lib, qs, ts, ks, ms, want_path):
    """The lane route must have taken every problem that fits it: a lane kernel that handed everything on would otherwise pass on the wave route's results."""
    lane, wave = api.myers_lanes_last_routes(lib)
    assert lane + wave == len(qs)
    must = sum(1 for q, t, k, m in zip(qs, ts, ks, ms) if _fits_lane_route(q, t, k, m, want_path) is True)
    cannot = sum(1 for q, t, k, m in zip(qs, ts, ks, ms) if _fits_lane_route(q, t, k, m, want_path) is False)
    assert lane >= must and wave >= cannot, (lane, wave, must, cannot)
    if not want_path:
        assert lane == must and wave == cannot, (lane, wave, must, cannot)
    else:  # with paths: everything whose whole table fits is the lane route's for sure
        sure = sum(1 for q, t, k, m in zip(qs, ts, ks, ms) if _fits_lane_route(q, t, k, m, False) is True and ((len(q) + 63) // 64) * len(t) <= 4096)
        assert lane >= sure, (lane, sure)


def _check(lib):
    rows = [r for r in golden_rows() if not r["path"] and set(r["q"] + r["t"]) <= set("ACGT")]
    assert len(rows) > 100
    res = api.myers_batch([r["q"] for r in rows], [r["t"] for r in rows], [r["k"] for r in rows], [r["mode"] for r in rows], use_iupac=False, lib_path=lib, lanes=True)
    _assert_routes(lib, [r["q"] for r in rows], [r["t"] for r in rows], [r["k"] for r in rows], [r["mode"] for r in rows], False)
    for r, (d, locs, _) in zip(rows, res):
        assert d == r["d"] and locs == r["locs"], (r["q"][:40], r["t"][:40], r["k"], r["mode"])
    qs, ts, ks, ms = _problems()
    res = api.myers_batch(qs, ts, ks, ms, use_iupac=False, lib_path=lib, lanes=True)
    _assert_routes(lib, qs, ts, ks, ms, False)
    assert api.myers_lanes_last_routes(lib)[1] > 0  # (and the hand-over is exercised: queries above 512 characters, IUPAC targets)
    for q, t, k, mode, (d, locs, _) in zip(qs, ts, ks, ms, res):
        w = op.myers(q, t, k, mode, False, iupac=False)
        assert (d, locs) == (w[0], w[1]), (len(q), len(t), k, mode)
    # paths: the golden vectors with a CIGAR (plain characters), then the random problems again (NW and SHW; HW paths are not on the hot path)
    rows = [r for r in golden_rows() if r["path"] and set(r["q"] + r["t"]) <= set("ACGT")]
    assert len(rows) > 50
    res = api.myers_batch([r["q"] for r in rows], [r["t"] for r in rows], [r["k"] for r in rows], [r["mode"] for r in rows], want_path=True, use_iupac=False, lib_path=lib, lanes=True)
    for r, (d, locs, cig) in zip(rows, res):
        assert d == r["d"] and locs == r["locs"] and cig == r["cigar"], (len(r["q"]), len(r["t"]), r["k"], r["mode"])
    sel = [i for i in range(len(qs)) if ms[i] != 2]
    res = api.myers_batch([qs[i] for i in sel], [ts[i] for i in sel], [ks[i] for i in sel], [ms[i] for i in sel], want_path=True, use_iupac=False, lib_path=lib, lanes=True)
    _assert_routes(lib, [qs[i] for i in sel], [ts[i] for i in sel], [ks[i] for i in sel], [ms[i] for i in sel], True)
    for i, (d, locs, cig) in zip(sel, res):
        w = op.myers(qs[i], ts[i], ks[i], ms[i], True, iupac=False)
        assert (d, locs, cig) == (w[0], w[1], w[2]), (len(qs[i]), len(ts[i]), ks[i], ms[i])
    # IUPAC equality (what the region program aligns with): every golden vector, distances and paths; random queries that carry codes against plain targets
    rows = [r for r in golden_rows() if not r["path"]]
    res = api.myers_batch([r["q"] for r in rows], [r["t"] for r in rows], [r["k"] for r in rows], [r["mode"] for r in rows], use_iupac=True, lib_path=lib, lanes=True)
    for r, (d, locs, _) in zip(rows, res):
        assert d == r["d"] and locs == r["locs"]
    rows = [r for r in golden_rows() if r["path"]]
    res = api.myers_batch([r["q"] for r in rows], [r["t"] for r in rows], [r["k"] for r in rows], [r["mode"] for r in rows], want_path=True, use_iupac=True, lib_path=lib, lanes=True)
    for r, (d, locs, cig) in zip(rows, res):
        assert d == r["d"] and locs == r["locs"] and cig == r["cigar"]
    rnd = random.Random(12)
    q2, t2, k2, m2 = [], [], [], []
    for i in range(300):
        m = rnd.choice((20, 64, 100, 130, 256, 400))
        t = "".join(rnd.choice("ACGT" if i % 3 else "ACGTN") for _ in range(m + rnd.randrange(-10, 40)))
        q = "".join((rnd.choice("MRSVWYHKDBN") if rnd.random() < 0.05 else c) for c in t[:m] if rnd.random() > 0.04)
        q2.append(q); t2.append(t); k2.append(-1 if i % 5 else 30); m2.append(i % 2)
    res = api.myers_batch(q2, t2, k2, m2, want_path=True, use_iupac=True, lib_path=lib, lanes=True)
    _assert_routes(lib, q2, t2, k2, m2, True)
    assert api.myers_lanes_last_routes(lib)[0] == len(q2)  # region-shaped problems: all of them one per lane
    for q, t, k, mode, (d, locs, cig) in zip(q2, t2, k2, m2, res):
        w = op.myers(q, t, k, mode, True, iupac=True)
        assert (d, locs, cig) == (w[0], w[1], w[2]), (len(q), len(t), k, mode)


def test_sim_myers_one_problem_per_lane():
    _check(SIM_LIB)


@pytest.mark.gpu
def test_gpu_myers_one_problem_per_lane():
    _check(None)
