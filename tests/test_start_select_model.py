"""not-gpu: the ALGORITHM of the per-start candidate table (device/start_select.hip) restated in numpy and checked against
the oracle's FindIter -- what the kernels compute, step by step, with small blocks so that a short haystack spans many:
  1. cand[i] = the one occurrence the leftmost rule can report from start i (LeftmostFirst: lowest id, LeftmostLongest:
     greatest length then lowest id);
  2. per block of B positions the function  entry offset o -> exit offset  of FindIter's chain pos -> end of the match
     chosen from pos (a match overshoots a block by less than the longest pattern L <= B);
  3. the chain across blocks is the COMPOSITION of those functions (here: a plain fold; on the device per group of 256 blocks
     in parallel, then over the groups);
  4. every block marks the candidates its part of the chain selects from its true entry offset.
The device code is checked against the oracle by tests/test_gpu_find_dense.py; this file pins the construction itself."""
import numpy as np
import pytest

from oracle import orc


def table_find_iter(pats, hay, rule, B, span=None, casei=False):
    lo, hi = span if span else (0, len(hay))
    occ = orc.Oracle(pats, ascii_case_insensitive=casei).find_overlapping_iter(hay, span=(lo, hi), as_numpy=True)
    L = max(map(len, pats))
    assert 1 <= L <= B
    n = hi - lo
    cand_pid = np.full(n, -1, dtype=np.int64)
    cand_len = np.zeros(n, dtype=np.int64)
    for p, s, e in zip(occ["pattern"].tolist(), occ["start"].tolist(), occ["end"].tolist()):
        i, ln = s - lo, e - s
        better = cand_pid[i] < 0 or (p < cand_pid[i] if rule == 1 else (ln > cand_len[i] or (ln == cand_len[i] and p < cand_pid[i])))
        if better:
            cand_pid[i], cand_len[i] = p, ln
    nblk = -(-n // B) if n else 0

    def block_chain(b, o):
        """positions of block b selected by a chain entering at offset o, and its exit offset"""
        pos, sel = b * B + o, []
        end = min((b + 1) * B, n)
        while True:
            nxt = next((q for q in range(pos, end) if cand_pid[q] >= 0), None)
            if nxt is None:
                return sel, max(pos - (b + 1) * B, 0) if pos >= (b + 1) * B else 0
            sel.append(nxt)
            pos = nxt + int(cand_len[nxt])
            if pos >= (b + 1) * B:
                return sel, pos - (b + 1) * B
    # step 2: e1[b][o] for every entry offset that can occur
    e1 = [[block_chain(b, o)[1] for o in range(min(L, B))] for b in range(nblk)]
    assert all(x < L for row in e1 for x in row)
    # step 3: composition -> the true entry offset of every block
    entry, o = [], 0
    for b in range(nblk):
        entry.append(o)
        o = e1[b][o]
    # step 4
    out = []
    for b in range(nblk):
        for q in block_chain(b, entry[b])[0]:
            out.append((int(cand_pid[q]), lo + q, lo + q + int(cand_len[q])))
    return out


@pytest.mark.parametrize("rule", [1, 2])
@pytest.mark.parametrize("seed", range(12))
def test_model_equals_find_iter(rule, seed):
    rng = np.random.default_rng(700 + seed)
    sigma = int(rng.integers(1, 4))
    pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(int(rng.integers(1, 12)))]
    n = int(rng.integers(0, 700))
    hay = rng.integers(0x61, 0x61 + sigma + int(rng.integers(0, 2)), size=n, dtype=np.uint8)
    o = orc.Oracle(pats, match_kind=rule)
    for B in (8, 16, 64):
        want = o.find_iter(hay, as_numpy=True)
        got = table_find_iter(pats, hay, rule, B)
        assert got == list(zip(want["pattern"].tolist(), want["start"].tolist(), want["end"].tolist())), (seed, rule, B)
    if n > 20:
        s, e = int(rng.integers(0, n // 2)), int(rng.integers(n // 2, n + 1))
        want = o.find_iter(hay, span=(s, e), as_numpy=True)
        assert table_find_iter(pats, hay, rule, 16, span=(s, e)) == list(zip(want["pattern"].tolist(), want["start"].tolist(), want["end"].tolist()))


def test_chains_that_never_merge():
    """'aa' over a run of 'a': the chains from even and from odd offsets stay apart for ever -- the per-block functions
    carry both, the composition picks the right one"""
    hay = np.full(301, 0x61, dtype=np.uint8)
    hay[100] = 0x62
    for pats in ([b"aa"], [b"aaa", b"a"], [b"aa", b"a"]):
        for rule in (1, 2):
            want = orc.Oracle(pats, match_kind=rule).find_iter(hay, as_numpy=True)
            assert table_find_iter(pats, hay, rule, 8) == list(zip(want["pattern"].tolist(), want["start"].tolist(), want["end"].tolist()))
