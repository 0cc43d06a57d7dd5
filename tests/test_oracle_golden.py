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
