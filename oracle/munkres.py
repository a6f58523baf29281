"""TEST INFRASTRUCTURE (oracle): the Hungarian / Munkres assignment the reference tracker calls.

`/root/reference/src/lib/utils/tracker.py:6,157` imports `sklearn.utils.linear_assignment_.linear_assignment`; the reference
pins scikit-learn==0.22.2 (requirements.txt:13).  That module was removed in scikit-learn 0.23 and is absent from this
image (1.7), and it is an un-vendored third-party dependency, so this file RESTATES its published algorithm -- the
Kuhn-Munkres state machine of `sklearn/utils/linear_assignment_.py` in 0.22.2 (steps 1, 3, 4, 5, 6 of the classic
description; identical to scipy <= 1.3's `optimize/_hungarian.py`, which was derived from it) -- operation for operation:
the same float64 subtractions and additions on the same elements in the same order, the same scan orders (row-major
`np.where`, first maximum of `argmax`), because WHICH optimum comes out among equal-cost assignments (the tracker's cost
matrices are full of 1e18 "forbidden" entries) depends on all of them, and the tracker hands out new ids in that order.

PARITY UNPINNED against the real module (it cannot be installed here, and the reference holds no vector for it).  What is
checked (tests/test_track_logic_cpu.py): its total cost equals scipy.optimize.linear_sum_assignment's on random, tie-heavy
and gate-heavy matrices of every aspect ratio; every row of the shorter side is assigned exactly once; the C++ restatement in
centerpose_amd/csrc/track_common.h (trk_munkres, device + host build) returns the same PAIRS as this file on all of them.
Only tests/ and oracle/tools/ import this module."""
import numpy as np


def _reduce_rows_and_star(C, marked):
    """Step 1 (+ 2): subtract each row's minimum; star a zero whose row and column hold no star yet, scanning the zeros in
    row-major order."""
    C -= C.min(axis=1)[:, np.newaxis]
    n, m = C.shape
    row_free = np.ones(n, bool)
    col_free = np.ones(m, bool)
    rows, cols = np.where(C == 0)
    for i, j in zip(rows.tolist(), cols.tolist()):
        if col_free[j] and row_free[i]:
            marked[i, j] = 1
            col_free[j] = False
            row_free[i] = False


def _prime_zeros(C, marked, row_unc, col_unc):
    """Step 4: prime uncovered zeros (the FIRST one in row-major order each time) until one has no star in its row
    (-> returns its position: augment) or none is left (-> returns None: adjust the matrix)."""
    n, m = C.shape
    Z = (C == 0).astype(np.int64)
    cov = Z * row_unc[:, np.newaxis].astype(np.int64)
    cov *= col_unc.astype(np.int64)
    while True:
        row, col = np.unravel_index(int(np.argmax(cov)), (n, m))
        if cov[row, col] == 0:
            return None
        marked[row, col] = 2
        star_col = int(np.argmax(marked[row] == 1))
        if marked[row, star_col] != 1:
            return int(row), int(col)
        col = star_col
        row_unc[row] = False
        col_unc[col] = True
        cov[:, col] = Z[:, col] * row_unc.astype(np.int64)
        cov[row] = 0


def _augment(marked, path, r0, c0):
    """Step 5: the alternating path primed zero -> star in its column -> prime in that star's row -> ...; stars on the path
    are removed, primes become stars; every prime is erased."""
    count = 0
    path[0] = (r0, c0)
    while True:
        col = path[count, 1]
        row = int(np.argmax(marked[:, col] == 1))
        if marked[row, col] != 1:
            break
        count += 1
        path[count] = (row, col)
        pcol = int(np.argmax(marked[row] == 2))
        if marked[row, pcol] != 2:
            pcol = -1
        count += 1
        path[count] = (row, pcol)
    for i in range(count + 1):
        r, c = path[i]
        marked[r, c] = 0 if marked[r, c] == 1 else 1
    marked[marked == 2] = 0


def _adjust(C, row_unc, col_unc):
    """Step 6: the smallest uncovered value is ADDED to every covered row, then SUBTRACTED from every uncovered column (an
    element of a covered row and an uncovered column sees both, in that order: two roundings)."""
    if row_unc.any() and col_unc.any():
        minval = np.min(C[row_unc], axis=0)
        minval = np.min(minval[col_unc])
        C[np.logical_not(row_unc)] += minval
        C[:, col_unc] -= minval


def linear_assignment(X):
    """-> int array [min(n, m), 2] of (row, column) pairs sorted by row (then column): sklearn 0.22.2's return value."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    if 0 in X.shape:
        return np.zeros((0, 2), dtype=int)
    transposed = X.shape[1] < X.shape[0]   # more rows than columns: work on the transpose, swap the pairs back
    C = (X.T if transposed else X).copy()
    n, m = C.shape
    marked = np.zeros((n, m), dtype=np.int64)
    path = np.zeros((n + m, 2), dtype=np.int64)
    _reduce_rows_and_star(C, marked)
    row_unc = np.ones(n, bool)
    col_unc = np.ones(m, bool)
    while True:
        # step 3: cover the starred columns; n stars = done
        stars = marked == 1
        col_unc[np.any(stars, axis=0)] = False
        if stars.sum() >= n:
            break
        while True:
            z = _prime_zeros(C, marked, row_unc, col_unc)
            if z is not None:
                break
            _adjust(C, row_unc, col_unc)
        _augment(marked, path, z[0], z[1])
        row_unc[:] = True
        col_unc[:] = True
    pairs = np.array(np.where(marked == 1)).T
    if transposed:
        pairs = pairs[:, ::-1]
    pairs = sorted(pairs.tolist())
    return np.array(pairs, dtype=int).reshape(-1, 2)
