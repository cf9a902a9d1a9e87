"""Comparing two per-iteration LM records (the CvLevMarq state machine, iteration by iteration).

A record row is (iteration, damping exponent of the first trial, exponent of the accepted trial, number of trial
steps, residual sum at the committed point, residual sum at the accepted point) -- the first six fields of both
ccalib_last_lm_trace and the oracle's orc_set_trace.

Two correct implementations of the same state machine take the same decisions as long as the decisions rest on
differences that are large against round-off.  Once an iteration changes the residual sum by less than 1e-12 of
itself, "did the error grow" compares sums that agree to the last bits and the accept/reject pattern is no longer
a property of the algorithm (the oracle's dense and Schur solvers part ways there too).  So: up to the first such
iteration the records must agree exactly in every decision and to 1e-9 relative in the sums; after it only the end
state is compared.
"""
import numpy as np

SETTLED = 1e-12


def settled_index(trace):
    """Index of the first iteration whose relative change of the residual sum is below SETTLED (len(trace) if none)."""
    t = np.asarray(trace)
    for k in range(t.shape[0]):
        if abs(t[k, 4] - t[k, 5]) < SETTLED * abs(t[k, 4]):
            return k
    return t.shape[0]


def assert_same_decisions(got, ref, what="", rel=1e-9):
    """`got` against the reference record `ref`: identical decisions, residual sums within `rel`, in every row up to
    the point where `ref` has settled.  Returns the number of rows compared."""
    got, ref = np.asarray(got)[:, :6], np.asarray(ref)[:, :6]
    k = settled_index(ref)
    assert k >= 1, "the reference run settles before its first iteration: nothing to compare"
    assert got.shape[0] >= k, f"{what}: run ended after {got.shape[0]} iterations, the reference settles at {k}"
    for i in range(k):
        assert got[i, 0] == ref[i, 0] == i
        assert tuple(got[i, 1:4]) == tuple(ref[i, 1:4]), (what, i, got[i, :4], ref[i, :4])
        assert abs(got[i, 4] - ref[i, 4]) <= rel * ref[i, 4], (what, i, got[i, 4], ref[i, 4])
        assert abs(got[i, 5] - ref[i, 5]) <= rel * ref[i, 5], (what, i, got[i, 5], ref[i, 5])
    return k
