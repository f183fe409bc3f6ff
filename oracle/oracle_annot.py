"""Second, independent restatement of the reference's INDEX-TIME annotators, for cross-checking the index producer
(`ratatosk_amd/csrc/tools/build_index.cpp`). TEST INFRASTRUCTURE ONLY (tests/test_annotators.py): nothing in the product path imports it.

The correction path reads three things an index carries besides unitigs and colours: the edge bits / branching bit
(src/Graph.cpp:1986-2021 `postProcessUnitigs`), the short-cycle strings (`detectShortCycles`, src/Graph.cpp:4660-4735) and the SNP
annotations (`detectSNPs`, src/Graph.cpp:484-573, with `isValidSNPcandidate`, src/GraphTraversal.cpp:1057-1147). The repo had ONE
implementation of these (the index tool, C++), read by oracle and product alike, so a coding error there was invisible. This module
recomputes all three from nothing but the unitig sequences and colour sets of a written index, in Python, with its own k-mer
dictionary and its own adjacency (built from the oriented head k-mers of the unitigs, not from the tool's table), and the test
compares them with what the index file holds, unitig by unitig.

Parity status: like the rest of oracle/ above the alignment layer this is **parity unpinned** against Bifrost (assumptions [A1]-[A3],
rule [D3] of oracle/oracle_graph.hpp:6-27 apply: getSuccessors() in A,C,G,T order of the appended base; the order of the 1-substitution
hits inside one window is (substituted offset, substituted base)).
"""
import numpy as np

_COMP = str.maketrans("ACGT", "TGCA")
_IUPAC = {0: ".", 1: "A", 2: "C", 3: "M", 4: "G", 5: "R", 6: "S", 7: "V", 8: "T", 9: "W", 10: "Y", 11: "H", 12: "K", 13: "D", 14: "B", 15: "N"}
_IUPAC_REV = {c: b for b, c in _IUPAC.items()}  # getAmbiguityRev / getAmbiguity (src/Common.hpp:351-399): bit 0 A, 1 C, 2 G, 3 T


def revcomp(s):
    return s.translate(_COMP)[::-1]


def _codes(seq):
    c = (np.frombuffer(seq.encode(), dtype=np.uint8).astype(np.uint64) >> np.uint64(1)) & np.uint64(3)  # A 0, C 1, T 2, G 3
    return c ^ (c >> np.uint64(1))  # A 0, C 1, G 2, T 3


def _windows(seq, k):
    """2-bit codes (first base in the high bits) of every k-mer of seq, as uint64 (k <= 31)."""
    c = _codes(seq)
    n = len(seq) - k + 1
    w = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        w = (w << np.uint64(2)) | c[j:j + n]
    return w


def _rc_codes(x, k):
    """Reverse complement of 2-bit k-mer codes (numpy uint64 array)."""
    y = ~x
    y = ((y >> np.uint64(2)) & np.uint64(0x3333333333333333)) | ((y & np.uint64(0x3333333333333333)) << np.uint64(2))
    y = ((y >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F)) | ((y & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4))
    y = y.byteswap()
    return y >> np.uint64(64 - 2 * k)


class AnnotGraph:
    """Unitigs + colour sets of an index, and the adjacency derived from them (nothing else is taken from the index)."""

    def __init__(self, seqs, colours, k, min_cov=2):
        self.seq, self.k, self.min_cov = list(seqs), k, min_cov
        self.col = [frozenset(c) for c in colours]
        self.heads = {}  # oriented head k-mer (text) -> (unitig, strand)
        for u, s in enumerate(self.seq):
            self.heads[s[:k]] = (u, True)
            self.heads[revcomp(s[-k:])] = (u, False)
        # every k-mer of the graph, canonical code -> unitig, for the 1-substitution search
        canon, uid = [], []
        for u, s in enumerate(self.seq):
            w = _windows(s, k)
            canon.append(np.minimum(w, _rc_codes(w, k)))
            uid.append(np.full(len(w), u, dtype=np.int64))
        canon, uid = np.concatenate(canon), np.concatenate(uid)
        o = np.argsort(canon, kind="stable")
        self._canon, self._uid = canon[o], uid[o]
        self._succ = {}

    def shares(self, a, b):
        """getNumberSharedPairID(a, b, min_cov) >= min_cov"""
        return len(self.col[a] & self.col[b]) >= self.min_cov

    def successors(self, u, fw):
        """getSuccessors() of unitig u read on strand fw: [(unitig, strand, appended base)] in A,C,G,T order ([A3])."""
        key = (u, fw)
        r = self._succ.get(key)
        if r is None:
            s = self.seq[u]
            end = s[-self.k:] if fw else revcomp(s[:self.k])
            r = []
            for b in "ACGT":
                h = self.heads.get(end[1:] + b)
                if h is not None:
                    r.append((h[0], h[1], b))
            self._succ[key] = r
        return r

    # ---- src/Graph.cpp:1986-2021 (postProcessUnitigs) + UnitigData.hpp:262-283: edge bits and the branching bit
    def edge_bits(self, u):
        bits = 0
        for fw in (True, False):
            for w, _, b in self.successors(u, fw):
                if self.shares(u, w):
                    idx = 1 << "ACGT".index(b)  # getAmbiguityIndex: A 1, C 2, G 4, T 8
                    bits |= (idx << 4) if fw else idx
        return bits

    def branching(self, u):
        return len(self.successors(u, True)) > 1 or len(self.successors(u, False)) > 1  # predecessors = successors of the other strand

    def has_edge(self, bits, fw, b):
        idx = 1 << "ACGT".index(b)
        return bool(bits & ((idx << 4) if fw else idx))

    # ---- src/Graph.cpp:4660-4735 (detectShortCycle(um, true)); Path::extend / getMiddleCompactedPath src/Path.hpp:307-330,805-808
    def short_cycles(self, u0, bits):
        """bits[u] = the low byte of shared_pids of every unitig. Returns the compact cycle strings in the order they are found."""
        k, out = self.k, []
        n_km0 = len(self.seq[u0]) - k + 1
        queue = [[(u0, True, "")]]  # a path = [(unitig, strand, base that entered it)]
        qi = 0
        while qi < len(queue):
            path = queue[qi]
            qi += 1
            cu, cfw, _ = path[-1]
            for w, wfw, b in self.successors(cu, cfw):
                if not self.has_edge(bits[cu], cfw, b):  # :4690 the edge is seen in enough reads
                    continue
                if not self.shares(cu, u0):  # :4693 still read-compatible with the start
                    continue
                if w == u0 and wfw:  # :4696 back on the start unitig, same strand
                    inner = [(x, f) for x, f, _ in path[1:]]
                    if len(set(inner)) != len(inner):  # :4706-4710 a smaller cycle inside
                        continue
                    pid = self.col[u0]
                    for x, _ in inner:  # :4716
                        if len(pid) < self.min_cov:
                            break
                        pid = pid & self.col[x]
                    if len(pid) >= self.min_cov:
                        out.append("".join(base for _, _, base in path[1:]))  # succ string: entering bases of the interior unitigs
                else:
                    length = k - 1 + sum(len(self.seq[x]) - k + 1 for x, _, _ in path)  # Path::length(): bases
                    if length - n_km0 < 2 * k:  # :4721 only short cycles are visited
                        queue.append(path + [(w, wfw, b)])
        return out

    # ---- src/GraphTraversal.cpp:1057-1147
    class _Walk:
        def __init__(self):
            self.m_km = {}  # (unitig, strand) [= mapped head k-mer] -> unitig whose colours are looked at
            self.q = []
            self.qi = 0

    def _explore(self, lgt, a, ub, bits, limit):
        ua, afw = a
        if len(self.col[ua]) < self.min_cov or len(self.col[ub]) < self.min_cov:
            return False
        if not lgt.m_km:
            lgt.q.append(a)
            lgt.m_km[a] = ua
        elif len(lgt.m_km) >= limit:
            return True
        while lgt.qi < len(lgt.q):
            xu, xfw = lgt.q[lgt.qi]
            lgt.qi += 1
            for w, wfw, b in self.successors(xu, xfw):
                if not self.has_edge(bits[xu], xfw, b):
                    continue
                if (w, wfw) in lgt.m_km:
                    continue
                lgt.m_km[(w, wfw)] = w
                if self.shares(w, ua):
                    if self.shares(w, ub):
                        return True
                    lgt.q.append((w, wfw))
            if len(lgt.m_km) >= limit:
                return True
        return False

    def _is_valid(self, fw, bw, ua, ub, bits, limit=65536):
        ok_fw = any(self.shares(x, ub) for x in fw.m_km.values()) or self._explore(fw, (ua, True), ub, bits, limit)
        if not ok_fw:
            return False
        return any(self.shares(x, ub) for x in bw.m_km.values()) or self._explore(bw, (ua, False), ub, bits, limit)

    # ---- src/Graph.cpp:484-573
    def snp_annotations(self, u, bits):
        """[(position, IUPAC code)] of unitig u."""
        if not bits[u] & 0xFF:  # hasSharedPids
            return []
        k, s = self.k, self.seq[u]
        w = _windows(s, k)
        n = len(w)
        # searchSequence(seq, exact = false, insertion = false, deletion = false, substitution = true, or_exclusive = false):
        # every graph k-mer one substitution away from a window; candidates by (window, offset, base)
        j = np.arange(k, dtype=np.uint64)
        sh = (np.uint64(2) * (np.uint64(k - 1) - j))[None, :, None]
        cur = (w[:, None, None] >> sh) & np.uint64(3)
        d = np.arange(1, 4, dtype=np.uint64)[None, None, :]
        alt = (cur + d) & np.uint64(3)
        var = (w[:, None, None] & ~(np.uint64(3) << sh)) | (alt << sh)
        flat = var.reshape(-1)
        can = np.minimum(flat, _rc_codes(flat, k))
        at = np.searchsorted(self._canon, can)
        at[at >= len(self._canon)] = 0
        hit = np.nonzero(self._canon[at] == can)[0]
        cand = []
        for h in hit:
            p, r = divmod(int(h), 3 * k)
            jj = r // 3
            a = int(alt.reshape(-1)[h])
            cand.append((p, jj, a, int(self._uid[at[h]])))
        cand.sort()
        final, tried = list(s), list(s)
        valid, invalid = set(), set()
        lgt_fw, lgt_bw = self._Walk(), self._Walk()
        for p, jj, a, wu in cand:
            if wu == u:  # :523 a SNP candidate cannot be on the same unitig
                continue
            pos = p + jj  # cstrMatch: first mismatch = the substituted offset
            kbit = 1 << a
            cf = _IUPAC[_IUPAC_REV[final[pos]] | kbit]
            ct = _IUPAC[_IUPAC_REV[tried[pos]] | kbit]
            if tried[pos] == ct:  # that base was tried at this position before
                continue
            tried[pos] = ct
            if wu in valid:
                final[pos] = cf
            elif wu not in invalid:
                if self._is_valid(lgt_fw, lgt_bw, u, wu, bits):
                    final[pos] = cf
                    valid.add(wu)
                else:
                    invalid.add(wu)
        return [(i, c) for i, c in enumerate(final) if c not in "ACGT"]


def from_oracle_graph(g, min_cov=2):
    """AnnotGraph over the unitigs and colour sets of an oracle_py.Graph (the loaded index), plus what the index says:
    (graph, shared words, kmcov words)."""
    seqs, cols, shared, kmcov = [], [], [], []
    globals_ = {}
    for u in range(g.n_unitigs):
        r = g.unitig(u)
        c = set(r["local"])
        if r["global_id"] >= 0:
            if r["global_id"] not in globals_:
                globals_[r["global_id"]] = set(g.global_set(r["global_id"]))
            c |= globals_[r["global_id"]]
        seqs.append(r["seq"])
        cols.append(c)
        shared.append(r["shared"])
        kmcov.append(r["kmcov"])
    return AnnotGraph(seqs, cols, g.k, min_cov), shared, kmcov
