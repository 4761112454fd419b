"""Library-format compatibility truth tables ported from the reference's own unit tests
(/root/reference/tests/LibraryTypeTests.cpp:11-183) — the one part of the hot path (SURVEY.md §8a row
a9) the reference pins.  Run-time formats follow src/util/LibraryTypeUtils.cpp:22-46 (ISF = SA,
ISR = AS); the reference test file spells ISF/ISR with S/A, which only matters for the
single-end/orphan table and is noted there."""
import itertools
import orc

T_SE, T_PE = 0, 1
O_SAME, O_AWAY, O_TOWARD, O_NONE = 0, 1, 2, 3
S_SA, S_AS, S_S, S_A, S_U = 0, 1, 2, 3, 4
MS_SE, MS_LEFT, MS_RIGHT = 0, 1, 2

FM = {"U": (T_SE, O_NONE, S_U), "SF": (T_SE, O_NONE, S_S), "SR": (T_SE, O_NONE, S_A),
      "IU": (T_PE, O_TOWARD, S_U), "ISF": (T_PE, O_TOWARD, S_SA), "ISR": (T_PE, O_TOWARD, S_AS),
      "OU": (T_PE, O_AWAY, S_U), "OSF": (T_PE, O_AWAY, S_SA), "OSR": (T_PE, O_AWAY, S_AS),
      "MU": (T_PE, O_SAME, S_U), "MSF": (T_PE, O_SAME, S_S), "MSR": (T_PE, O_SAME, S_A)}


def test_format_id_roundtrip(built):
    L = orc.lib()
    ids = {n: L.orc_format_id(*f) for n, f in FM.items()}
    assert len(set(ids.values())) == len(ids)          # LibraryTypeTests.cpp:11-37: encode/decode is injective
    for n, (t, o, s) in FM.items():
        i = ids[n]
        assert (i & 1, (i >> 1) & 3, i >> 3) == (t, o, s)


def test_paired_end_compatibility_table(built):
    # LibraryTypeTests.cpp:40-92
    L = orc.lib()
    for en, on in itertools.product(FM, ["ISF", "ISR", "OSF", "OSR", "MSF", "MSR"]):
        got = bool(L.orc_compatible_pe(*FM[en], *FM[on]))
        want = (en == on) or (en == "IU" and on in ("ISF", "ISR")) or (en == "OU" and on in ("OSF", "OSR")) or \
               (en == "MU" and on in ("MSF", "MSR"))
        assert got == want, (en, on)


def test_single_end_and_orphan_compatibility_table(built):
    # LibraryTypeTests.cpp:96-183, with "strandedness S" read as "read 1 from the sense strand"
    # (SA for I/O libraries, S for M/SE) and "A" as antisense (AS / A).
    L = orc.lib()
    for en, (t, o, s) in FM.items():
        for fwd, ms in itertools.product([True, False], [MS_LEFT, MS_RIGHT, MS_SE]):
            if t == T_SE and ms in (MS_LEFT, MS_RIGHT):
                continue
            if t == T_PE and ms == MS_SE:
                continue
            got = bool(L.orc_compatible_se(t, o, s, int(fwd), ms))
            sense = s in (S_SA, S_S); anti = s in (S_AS, S_A)
            if s == S_U:
                want = True
            elif sense and o != O_SAME:
                want = (fwd and ms == MS_SE) or (fwd and ms == MS_LEFT) or (not fwd and ms == MS_RIGHT)
            elif anti and o != O_SAME:
                want = (not fwd and ms == MS_SE) or (not fwd and ms == MS_LEFT) or (fwd and ms == MS_RIGHT)
            else:
                want = (sense and fwd) or (anti and not fwd)
            assert got == want, (en, fwd, ms)
