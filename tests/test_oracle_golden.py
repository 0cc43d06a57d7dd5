"""Pin the oracle's loss family against the reference's own NumPy oracles
(model/test_utils.py:157-318) through tests/golden/loss_golden.npz, with the
tolerance the reference's self-tests use (np.allclose defaults: rtol 1e-5,
atol 1e-8 - model/tdnn.py:283,314,343)."""
import os

import numpy as np
import pytest

from oracle import xvector_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))


def _oracle_loss(i, dtype):
    kind = str(G["kind"][i])
    m = float(G["m"][i])
    if kind == "asoftmax":
        m = int(m)
    lmin, lbase, lgamma, lpower = G["sched"][i]
    emb = G["emb"][i].astype(dtype)
    w = G["w"][i].astype(dtype)
    if G["feature_norm"][i]:
        emb, _ = O.l2_scaling_fwd(emb, 0.1)
    lam = O.margin_lambda(lmin, lbase, lgamma, lpower, int(G["step"][i]))
    loss, _, grads = O.margin_softmax_loss(kind, emb, G["labels"][i], w, m, lam)
    return loss, grads


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_loss_family_matches_reference_numpy_oracles(dtype):
    n = len(G["loss"])
    assert n >= 100
    worst = 0.0
    for i in range(n):
        loss, grads = _oracle_loss(i, dtype)
        ref = G["loss"][i]
        # the reference's bar: np.allclose(loss_tf, loss_np)
        tol = 1e-8 + 1e-5 * abs(ref) if dtype == np.float64 else 1e-6 + 2e-5 * abs(ref)
        assert abs(float(loss) - ref) <= tol, (i, str(G["kind"][i]), G["m"][i], float(loss), ref)
        worst = max(worst, abs(float(loss) - ref) / max(abs(ref), 1e-12))
        for g in grads:
            assert np.all(np.isfinite(g)), "Gradient should not be nan (model/tdnn.py:282)"
    assert worst < 5e-5


def test_lambda_schedule_values():
    # shipped AM schedule: min 0, base 1000, gamma 1e-4, power 5 (tdnn_amsoftmax_*.json:9-13)
    assert O.margin_lambda(0, 1000, 1e-4, 5, 0) == 1000.0
    assert abs(O.margin_lambda(0, 1000, 1e-4, 5, 10000) - 1000 * 2.0 ** -5) < 1e-12
    assert O.margin_lambda(10, 1000, 1e-5, 5, 10 ** 9) == 10.0


def test_self_attention_matches_reference_numpy_oracle():
    """oracle.self_attention_fwd vs model/test_utils.py:compute_self_attention (tests/golden/make_attention_golden.py).
    The reference oracle adds 1e-12 inside the sqrt where the graph (and our restatement) masks variances <= 1e-12
    (pooling.py:160-162): identical above ~1e-9, both give 1e-6 for a constant chunk."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "attention_golden.npz"))
    for i in range(int(g["num_cases"])):
        value, key, query, ref = g["value_%d" % i], g["key_%d" % i], g["query_%d" % i], g["att_%d" % i]
        att, _ = O.self_attention_fwd(value, key, query, bool(g["use_scale_%d" % i]))
        assert att.shape == ref.shape
        assert np.allclose(att, ref, rtol=1e-5, atol=1e-8), (i, np.abs(att - ref).max())


def test_auxiliary_losses_match_reference_numpy_oracles():
    """oracle.ring_loss / mhe_loss vs model/test_utils.py:855-884 (tests/golden/make_aux_golden.py).  The graph (and the
    restatement) adds 1e-6 to the mean distance of MHE (loss.py:1028), the NumPy oracle does not: 1e-6 relative."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "aux_golden.npz"))
    for i in range(int(g["num_cases"])):
        lam = float(g["lambda_%d" % i])
        ring, _ = O.ring_loss(g["features_%d" % i], float(g["r_%d" % i]), lam)
        assert np.isclose(ring, float(g["ring_%d" % i]), rtol=1e-12)
        mhe, _ = O.mhe_loss(g["w_%d" % i], g["labels_%d" % i], lam)
        assert np.isclose(mhe, float(g["mhe_%d" % i]), rtol=5e-6), (mhe, float(g["mhe_%d" % i]))


def _pooling_golden_cases():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from pooling_cases import pooling_cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pooling_golden.npz"))
    xs = list(pooling_cases())
    assert len(xs) == int(g["num_cases"])
    for i, x in enumerate(xs):
        x64 = x.astype(np.float64)
        # the regenerated input is the one the reference function saw
        assert np.allclose([x64.sum(), np.abs(x64).sum()], g["x_sum_%d" % i], rtol=1e-13, atol=0)
        yield i, x, g["pool_%d" % i]


def check_pooling_against_reference(got, ref, x, rel_mean, rel_std, name):
    """Shared by the oracle test and the GPU test.  The reference function returns sqrt(var + 1e-12) where the graph (pooling.py:28-29)
    returns sqrt(var) for var > 1e-12 and 1e-6 otherwise: identical at a constant chunk (1e-6), 5e-13 / var relative elsewhere.  The
    x 1e-8 row has var ~ 8e-18, far below the mask threshold, so both say 1e-6 there too."""
    c = x.shape[2]
    scale = np.abs(x.astype(np.float64)).max(axis=(1, 2))[:, None] + 1e-30
    assert got.shape == ref.shape, name
    assert np.all(np.isfinite(got)), name
    assert np.all(np.abs(got[:, :c] - ref[:, :c]) <= rel_mean * scale), (name, np.abs(got[:, :c] - ref[:, :c]).max())
    var_ref = ref[:, c:] ** 2 - 1e-12
    floor = var_ref <= 1e-9                     # constant / zero / x 1e-8 rows and T = 1: the epsilon decides
    assert np.allclose(got[:, c:][floor], 1e-6, rtol=1e-3, atol=0), name
    rest = ~floor
    err = np.abs(got[:, c:] - ref[:, c:])
    assert np.all(err[rest] <= (rel_std * np.broadcast_to(scale, err.shape))[rest] + 1e-3 * 1e-6), (name, err[rest].max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_statistics_pooling_matches_reference_numpy(dtype):
    """oracle.statistics_pooling_fwd (model/pooling.py:9-34) vs the reference's compute_self_attention with a zero query = uniform
    weights (tests/golden/make_pooling_golden.py): the one frame-level row of SURVEY 8(a) the reference's own code can pin."""
    n = 0
    for i, x, ref in _pooling_golden_cases():
        got, _ = O.statistics_pooling_fwd(x.astype(dtype))
        tol = (1e-12, 1e-9) if dtype == np.float64 else (2e-6, 2e-5)
        check_pooling_against_reference(np.asarray(got, np.float64), ref, x, tol[0], tol[1], "case %d" % i)
        n += 1
    assert n == 5
