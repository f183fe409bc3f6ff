"""Generates tests/golden/edlib_golden.tsv by running the REFERENCE's own edlib (oracle/_ref/libedlib_ref.so, compiled by
oracle/Makefile from /root/reference/src/edlib.cpp where it lies). Run in the build container only; the TSV (data, not
source) is committed so that the GPU box -- which has no /root/reference -- can pin both the oracle and the HIP kernels.

Columns: query, target, k, mode(0 NW,1 SHW,2 HW), want_path, editDistance, endLocations(comma sep or '-'), cigar(or '-').
Coverage follows SURVEY.md §8c: lengths {0,1,31,63,64,65,127,128,129,500,1031,1300}, error rates {0,1,5,10,25,60 %},
IUPAC letters, k in {-1, exact, exact-1, 0}, cases on both sides of the 1 MB traceback/Hirschberg switch.
"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle_py as op  # noqa: E402


def main():
    rnd = random.Random(20250928)
    rs = lambda n, a="ACGT": "".join(rnd.choice(a) for _ in range(n))

    def mut(s, e):
        o = []
        for c in s:
            if rnd.random() < e:
                r = rnd.randrange(3)
                if r == 0:
                    o.append(rnd.choice("ACGT"))
                elif r == 1:
                    o.append(c); o.append(rnd.choice("ACGT"))
            else:
                o.append(c)
        return "".join(o)

    rows = []

    def emit(q, t, k, mode, path):
        d, locs, cig = op.ref_edlib(q, t, k, mode, path)
        rows.append("\t".join([q or "-", t or "-", str(k), str(mode), str(int(path)), str(d), ",".join(map(str, locs)) or "-", cig or "-"]))
        return d

    lens = [0, 1, 31, 63, 64, 65, 127, 128, 129, 500, 1031, 1300]
    errs = [0.0, 0.01, 0.05, 0.10, 0.25, 0.60]
    for m in lens:
        for e in errs:
            q = rs(m)
            t = mut(q, e)
            if rnd.random() < 0.3:
                t = rs(rnd.randrange(40)) + t + rs(rnd.randrange(40))
            if rnd.random() < 0.25:
                t = "".join(c if rnd.random() > 0.04 else rnd.choice("NRYKMSWBDHV") for c in t)
            if rnd.random() < 0.15 and q:
                q = "".join(c if rnd.random() > 0.03 else "N" for c in q)
            for mode in (0, 1, 2):
                d = emit(q, t, -1, mode, False)
                if d >= 0:
                    emit(q, t, d, mode, False)
                    if d > 0:
                        emit(q, t, d - 1, mode, False)
                    emit(q, t, 0, mode, False)
                if mode < 2 and len(q) <= 500:
                    emit(q, t, -1, mode, True)
    # all-mismatch corner (pseudo position -1) and tiny cases
    for q, t in [("A", "C"), ("AAAA", "CCCC"), ("A" * 64, "C" * 70), ("A" * 65, "C" * 70), ("ACGT", "ACGT"), ("N", "A"), ("R", "N"), ("ACGTN", "ACGTA")]:
        for mode in (0, 1, 2):
            emit(q, t, -1, mode, False)
            if mode < 2:
                emit(q, t, -1, mode, True)
    # both sides of the 1 MB traceback / Hirschberg switch (20*ceil(q/64)*t + 8*t >= 2^20)
    for (m, n) in [(1300, 2480), (1300, 2520), (3300, 1000), (3400, 1000), (6000, 300)]:
        q = rs(m); t = mut(q, 0.08)[:n] if n <= m else mut(q + rs(n - m), 0.08)
        emit(q, t, -1, 0, True)
        emit(q, t, -1, 1, True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "edlib_golden.tsv")
    with open(out, "w") as f:
        f.write("\n".join(rows) + "\n")
    print("wrote", len(rows), "vectors to", out)


if __name__ == "__main__":
    main()
