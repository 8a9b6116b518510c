"""CPU property test of the exact parallel evaluation of sequential sums (csrc/mlx_seqfold.h, round 6).

The reference's reductions are sequential loops (bw/Tron.java:204-252, llf/LogisticRegressionL2.java:172-189); the reference-order
contract must reproduce their bits. mlx_seqfold.h does so without the dependency chain: inside a binade of the running sum the chain is
the exact sum of grid-rounded terms; where a prefix leaves the binade, a term ties or is too large, the literal chain takes over for
one sub-block. The HOST MODEL in that header makes the same decisions with the same arithmetic as the wave code (the GPU run of the
wave code itself: tools/seqfold_selftest.hip, pytest -m gpu). Here: >= 10^6 vectors of 15 kinds against `for (i) s += t[i]`."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "seqfold_host")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "ml-ease_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "seqfold_host.cpp"), "-o", exe], check=True, timeout=300)
    return exe


def _run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TOTAL mismatches 0" in r.stdout, r.stdout[-3000:] + r.stderr[-500:]
    rows = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("kind"):
            f = ln.split()
            kind = int(f[1])
            g = lambda key: int(f[f.index(key) + 1])       # noqa: E731
            rows[kind] = dict(vectors=g("vectors"), mismatches=g("mismatches"), blocks=g("sub-blocks"), failed=g("failed-checks"), literal=g("literal"))
    return rows


def test_a_million_random_and_adversarial_vectors_fold_to_the_sequential_bits(tmp_path):
    """15 kinds x 70 000 short vectors (1..300 terms: every start-up regime -- zero sums, the first binades, padding) + 15 x 30 long ones
    (20 000..80 000 terms, the step kernel's sizes): positive terms, random walks, wide dynamic range, few-mantissa-bit terms and sums in
    [2^53, 2^54) (every term a tie), alternating cancellation, powers of two, drifts with rare large steps, near-subnormal terms, one
    huge term, signed zeros, NaN / Inf inside, sign flips through zero, exact hits of binade edges. Zero mismatches."""
    exe = _build(tmp_path)
    rows = _run(exe, 70000, 30, 16, 1)
    assert len(rows) == 15 and sum(r["vectors"] for r in rows.values()) >= 1_000_000
    assert all(r["mismatches"] == 0 for r in rows.values())


def test_other_sub_block_lengths_and_what_the_grid_saves(tmp_path):
    """The result does not depend on the sub-block length K (8, 32); and on the sums the step kernel actually folds -- positive terms
    (dots of a vector with itself, norms, the loss) -- under 4 % of the sub-blocks of a long vector take the literal chain, the rest
    are added by the scan: that is where k_ro_step's time went."""
    exe = _build(tmp_path)
    for K in (8, 32):
        rows = _run(exe, 3000, 4, K, 5)
        assert all(r["mismatches"] == 0 for r in rows.values())
    rows = _run(exe, 0, 60, 16, 9)
    for kind in (0, 2, 13):                                   # squares, wide dynamic range, squares behind a large start
        assert rows[kind]["literal"] <= 0.04 * rows[kind]["blocks"], (kind, rows[kind])
    assert rows[13]["failed"] == 0
