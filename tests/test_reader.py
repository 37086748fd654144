"""The SDPA reader's own conversions (csrc/host/problem.c: lines scanned in place, Clinger's exact short path for decimal values,
radix sorts for the entry order and the union pattern) against the plain ones they replace (sscanf per line, strtod, qsort --
LORADS_READER=sscanf) and against Python's correctly rounded float(): the same problem bit for bit.  Reference reader:
io/lorads_file_io.c:21-293 (fscanf)."""
import ctypes as C
import glob
import os
import random
import struct

import numpy as np
import pytest

from lorads_amd import host, instances
from tests import common


def _parse(lib, line):
    ij = (C.c_int * 4)()
    val = C.c_double(0.0)
    lib.lrd_parse_entry_line.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.lrd_parse_entry_line.restype = C.c_int
    n = lib.lrd_parse_entry_line(line.encode(), ij, C.byref(val))
    return n, list(ij), val.value


def _bits(x):
    return struct.pack("<d", x)


def _random_value(rng):
    kind = rng.randrange(9)
    if kind == 0:
        return repr(rng.gauss(0, 1))                                  # 16-17 digits: strtod's turn
    if kind == 1:
        return "%.6f" % rng.uniform(-1000, 1000)
    if kind == 2:
        return "%d" % rng.randrange(-10**6, 10**6)
    if kind == 3:
        return "%.*e" % (rng.randrange(0, 18), rng.uniform(-1, 1) * 10.0 ** rng.randrange(-30, 30))
    if kind == 4:
        return "%s%d.%0*d" % (rng.choice(["", "+", "-"]), rng.randrange(0, 10**rng.randrange(1, 10)), rng.randrange(1, 12), rng.randrange(0, 10**9))
    if kind == 5:                                                     # mantissas around 2^53, exponents around +-22
        m = (1 << 53) + rng.randrange(-3, 4)
        return "%d%s" % (m, rng.choice(["", "e0", "e-22", "e22", "e23", "e-23", "E+5", "e-5"]))
    if kind == 6:
        return rng.choice(["0", "-0", "0.0", ".5", "5.", "-.25", "1e", "1e+", "1.e3", "0x10", "0x1p3", "inf", "-inf", "nan", "1d5", "1.5.3",
                           "00012.5000", "0.000000000000000000001", "123456789012345678901234567890", "1e400", "-1e-400", "4.9e-324"])
    if kind == 7:
        return "%.17g" % (rng.uniform(-1, 1) * 10.0 ** rng.randrange(-300, 300))
    return "%.3g" % rng.uniform(-1e5, 1e5)


def test_entry_line_values_are_strtods_bit_for_bit():
    lib = host.host_lib()
    rng = random.Random(925)
    fast = 0
    for _ in range(40000):
        tok = _random_value(rng)
        sep = rng.choice([" ", "  ", "\t", " \t "])
        ints = [rng.randrange(0, 10**rng.randrange(1, 7)) for _ in range(4)]
        line = rng.choice(["", " ", "\t"]) + sep.join(str(i) for i in ints) + sep + tok + rng.choice(["", " ", "\r", " trailing words"])
        n, ij, val = _parse(lib, line)
        try:
            # what sscanf("%lg") takes: the longest prefix strtod accepts (Python's float() wants the whole token)
            want = None
            for end in range(len(tok), 0, -1):
                t = tok[:end]
                if t.lower().startswith(("0x", "-0x", "+0x")):
                    try:
                        want = float.fromhex(t)
                        break
                    except ValueError:
                        continue
                if "_" in t or t.strip() != t:
                    continue
                try:
                    want = float(t)
                    break
                except ValueError:
                    continue
        except OverflowError:
            want = None
        if want is None:
            assert n == 4, (line, n)
            continue
        assert n == 5 and ij == ints, (line, n, ij)
        if want != want:
            assert val != val, (line, val)
        else:
            assert _bits(val) == _bits(want), (line, val, want)
        fast += 1
    assert fast > 35000


@pytest.mark.parametrize("line,nfields", [("", 0), ("   ", 0), ("1 2 3", 3), ("1 2 3 4", 4), ("1 2 3 4 x", 4), ("1,2,3,4,5.0", 1), ("1.5 2 3 4 5", 1),
                                          ("* comment", 0), ("1 2 3 4 5", 5), ("-1 +2 3 4 -5e-1", 5), ("1 2 3 99999999999 1.0", 3)])
def test_entry_line_field_counts_are_sscanfs(line, nfields):
    n, ij, val = _parse(host.host_lib(), line)
    assert n == nfields, (line, n)


def _digest(path, mode):
    lib = host.host_lib()
    lib.lrd_problem_digest.argtypes = [C.c_void_p]
    lib.lrd_problem_digest.restype = C.c_uint64
    old = os.environ.get("LORADS_READER")
    os.environ["LORADS_READER"] = mode
    try:
        s = host.Session.open(path)
    finally:
        if old is None:
            os.environ.pop("LORADS_READER")
        else:
            os.environ["LORADS_READER"] = old
    try:
        lib.lrd_session_problem.restype = C.c_void_p
        s.set_params(verbose=0)
        s.prepare(1, 0)
        return int(lib.lrd_problem_digest(s.problem_ptr())), s.m, s.nblk
    finally:
        s.close()


GOLDEN_FILES = sorted(glob.glob(os.path.join(common.GOLD, "*.dat-s")))


@pytest.mark.parametrize("path", GOLDEN_FILES, ids=[os.path.basename(p) for p in GOLDEN_FILES])
def test_same_problem_as_the_plain_reader_on_every_golden_file(path):
    assert _digest(path, "fast") == _digest(path, "sscanf")


def test_same_problem_on_an_untidy_file(tmp_path):
    """shuffled entry order, duplicates, upper-triangle entries, tiny values, CRLF line ends, comment lines in front, braces and commas
    in the header, a trailing comment section, no newline at the end"""
    rng = random.Random(7)
    lines = ['"a comment line"', "* another", " 7 = m", "3 = blocks", "{5, 4, -6}", "{1.0, -2.5e0, 3, 4.25, 0.125, 1e-1, 7}"]
    ents = []
    for _ in range(400):
        blk = rng.randrange(1, 4)
        n = [5, 4, 6][blk - 1]
        i = rng.randrange(1, n + 1)
        j = i if blk == 3 else rng.randrange(1, n + 1)
        ents.append("%d %d %d %d %s" % (rng.randrange(0, 8), blk, i, j, rng.choice([repr(rng.gauss(0, 1)), "%.4f" % rng.uniform(-3, 3), "1e-13", "2", "-0.5e1"])))
    lines += ents + ["", "   ", "end of data: what follows is not read", "1 1 1 1 5.0"]
    path = str(tmp_path / "untidy.dat-s")
    with open(path, "wb") as f:
        f.write("\r\n".join(lines).encode())
    a, b = _digest(path, "fast"), _digest(path, "sscanf")
    assert a == b and a[1] == 7 and a[2] == 3


def test_same_problem_on_a_large_generated_file(tmp_path):
    """10^5 entries with 17-digit values (strtod's path) and short ones (the exact short path), many cones: the radix sorts against qsort"""
    path = str(tmp_path / "mix.dat-s")
    instances.write_sdpa(instances.NAMED["mix4"](), path)
    assert _digest(path, "fast") == _digest(path, "sscanf")
    big = str(tmp_path / "rand4000.dat-s")
    instances.write_sdpa(instances.NAMED["rand4000"](), big)
    assert _digest(big, "fast") == _digest(big, "sscanf")


def test_start_point_generator_is_the_c_librarys(monkeypatch):
    """lrd_init_point draws the reference's start point (srand(925); two rand() per element, data/lorads_solver.c:361-371) from an inline
    copy of glibc's recurrence; the same numbers as rand() itself, element by element, on a multi-cone instance with an LP block"""
    lib = host.host_lib()
    assert lib.lrd_start_generator_is_inline() == 1
    pts = {}
    for mode in ("inline", "libc"):
        if mode == "libc":
            monkeypatch.setenv("LORADS_LIBC_RAND", "1")
        s = host.Session.open(common.instance_path("sdplp40"))
        s.set_params(verbose=0)
        s.prepare(1, 0)
        pts[mode] = [s.start_point(w, k) for k in range(s.nblk) for w in range(3)]
        s.close()
    assert len(pts["inline"]) >= 6
    for a, b in zip(pts["inline"], pts["libc"]):
        assert a.shape == b.shape and np.array_equal(a, b)
    s = host.Session.open(common.instance_path("maxcut800"))
    s.set_params(verbose=0, timesLogRank=6.0)
    s.prepare(1, 0)
    monkeypatch.delenv("LORADS_LIBC_RAND")
    t = host.Session.open(common.instance_path("maxcut800"))
    t.set_params(verbose=0, timesLogRank=6.0)
    t.prepare(1, 0)
    for w in range(3):
        assert np.array_equal(s.start_point(w, 0), t.start_point(w, 0))
    s.close(); t.close()
