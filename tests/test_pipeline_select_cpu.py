"""Which pipeline an MSM takes -- accumulation_amd/csrc/msm_select.h, the ONE table-driven function behind every entry point --
at every threshold edge, without a GPU: tests/cpp_host/select_check.cpp is plain C++ over the header alone.
Edges: n = 2^k - 1, 2^k, 2^k + 1 for k = 15 .. 20, 2^21, 2^22; keys: a direct-sum key, 16-bit tables of 2^16 / 2^19 generators,
the 20-bit table (2^20 and 2^22 generators), a plain key; forms: plain, grouped (regular / irregular index classes), skewed."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp_host", "select_check.cpp")
EXE = os.path.join(ROOT, "build", "select_check")
P = lambda k: 1 << k  # noqa: E731


@pytest.fixture(scope="module")
def table():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", SRC, "-o", EXE])
    out = subprocess.run([EXE], capture_output=True, text=True, check=True).stdout.splitlines()
    rows, other = {}, []
    for ln in out:
        w = ln.split()
        if len(w) == 8 and w[4].startswith("twin="):
            rows[(w[0], int(w[1]), w[2])] = {"pipeline": w[3], "twin": int(w[4][5:]), "plain_window": int(w[5].split("=")[1]),
                                            "range": int(w[6].split("=")[1]), "probe": int(w[7].split("=")[1])}
        else:
            other.append(ln)
    return rows, other


def expect(table, key, n, form, pipeline, twin=0, plain_window=0, rng=0):
    got = table[0][(key, n, form)]
    assert (got["pipeline"], got["twin"], got["plain_window"], got["range"]) == (pipeline, twin, plain_window, rng), (key, n, form, got)


def test_direct_sum_keys(table):
    for n in (1, P(15) - 1, P(15)):
        expect(table, "direct_2p15", n, "plain", "direct_sum")
        expect(table, "direct_2p15", n, "grouped", "direct_sum")
        expect(table, "direct_2p15", n, "skewed", "direct_sum")  # no buckets: nothing to skew
        expect(table, "direct_2p15", n, "grouped_irregular", "chunked")  # index classes of unequal size: the windowed pipelines


@pytest.mark.parametrize("key", ["table_2p16", "table_2p19"])
def test_precomputed_keys_without_a_20_bit_table(table, key):
    top = P(16) if key == "table_2p16" else P(19)
    for n, want in ((1, "chunked"), (P(15), "chunked"), (P(15) + 1, "chunked"), (P(16) - 1, "chunked"), (P(16), "bucket_split"),
                    (P(16) + 1, "bucket_split"), (P(17) - 1, "bucket_split"), (P(17), "bucket_split"), (P(17) + 1, "chunked"),
                    (P(18), "chunked"), (P(19), "chunked")):
        if n > top:
            continue
        for form in ("plain", "grouped", "grouped_irregular"):
            expect(table, key, n, form, want)
        expect(table, key, n, "skewed", "chunked")
        assert table[0][(key, n, "plain")]["probe"] == (1 if want == "bucket_split" else 0)


@pytest.mark.parametrize("key", ["bpl_2p20", "bpl_2p22"])
def test_keys_with_the_20_bit_table(table, key):
    for n, want, twin in ((1, "chunked", 1), (P(15), "chunked", 1), (P(16) - 1, "chunked", 1), (P(16), "bucket_split", 1), (P(17), "bucket_split", 1),
                          (P(17) + 1, "chunked", 1), (P(18) - 1, "chunked", 1), (P(18), "chunked", 1), (P(18) + 1, "bucket_per_lane", 0),
                          (P(19), "bucket_per_lane", 0), (P(19) + 1, "bucket_per_lane", 0), (P(20) - 1, "bucket_per_lane", 0),
                          (P(20), "bucket_per_lane", 0)):
        for form in ("plain", "grouped", "grouped_irregular"):
            expect(table, key, n, form, want, twin)
        expect(table, key, n, "skewed", "chunked", 1)  # a skewed vector: chunked over the 17-bit twin
    # longer than the window: ranges of 2^20 (which share one bucket set)
    for n in ((P(20) + 1, P(21), P(22) - 1, P(22)) if key == "bpl_2p22" else ()):
        for form in ("plain", "skewed"):
            expect(table, key, n, form, "chunked", 0, 0, P(20))
        assert table[0][(key, n, "plain")]["probe"] == 1


def test_plain_keys(table):
    key = "plain_2p22"
    for n, want, c in ((1, "chunked", 0), (P(15), "chunked", 0), (P(16), "chunked", 0), (P(17) - 1, "chunked", 0), (P(17), "chunked", 0),
                       (P(17) + 1, "bucket_per_lane", 15), (P(18), "bucket_per_lane", 15), (P(18) + 1, "bucket_per_lane", 16),
                       (P(19), "bucket_per_lane", 16), (P(19) + 1, "bucket_per_lane", 16), (P(20), "bucket_per_lane", 16)):
        for form in ("plain", "grouped", "grouped_irregular"):
            expect(table, key, n, form, want, 0, c)
        expect(table, key, n, "skewed", "chunked")
        assert table[0][(key, n, "plain")]["probe"] == 0  # plain keys find out from the prep's overflow flag
    for n in (P(20) + 1, P(21), P(22)):
        expect(table, key, n, "plain", "chunked", 0, 0, P(20))


def test_switches_and_key_windows(table):
    other = table[1]
    assert "switch bpl=0 bpl_2p20 chunked" in other
    assert "switch bpl_plain=0 plain chunked range=0" in other
    assert "switch bps=0 table_2p16 chunked" in other
    assert "switch bps=1 table_2p16 plain chunked grouped bucket_split" in other
    assert "switch direct=0 direct_2p15 chunked" in other
    assert "switch window_override bpl_2p20 chunked twin=1 plain chunked" in other
    assert "split table_2p23 default range=2097152 off range=0 below range=0" in other
    win = {ln.split()[1]: [int(x.split("=")[1]) for x in ln.split()[2:]] for ln in other if ln.startswith("key_window")}
    # (precomputed, precomputed with AMSM_BPL=0, plain on the chunked pipeline)
    assert win == {"2p1": [8, 8, 4], "2p8": [8, 8, 4], "2p9": [8, 8, 8], "2p12": [8, 8, 8], "2p13": [10, 10, 8], "2p14": [10, 10, 8],
                   "2p15": [13, 13, 8], "2p16": [16, 16, 8], "2p17": [17, 17, 8], "2p18": [16, 16, 8], "2p19": [16, 16, 8],
                   "2p20": [20, 17, 13], "2p22": [20, 17, 16]}
