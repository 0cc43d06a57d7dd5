"""The scheduled update (xv_engine_backward_update, include/xvector_hip.h): backward pass and optimiser step as one pass whose update and
weight-copy launches are enqueued per backward stage.  It is a SCHEDULE, not new arithmetic: everything it leaves behind must equal
xv_engine_backward + xv_engine_apply bit for bit.  Also here: the arena of a predict-only engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


SCHEDULED_CASES = [
    dict(kw=dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True), B=6, T=40),
    dict(kw=dict(loss_func="softmax", optimizer="momentum", momentum=0.9, use_nesterov=True), B=5, T=33),
    dict(kw=dict(loss_func="softmax", optimizer="adam"), B=4, T=30),
    dict(kw=dict(loss_func="additive_margin_softmax", margin_m=0.2, last_layer_linear=True, pooling_type="self_attention",
                 att_key_num_nodes=(64, 48)), B=4, T=30),
    dict(kw=dict(loss_func="asoftmax", margin_m=2, lambda_min=0, lambda_gamma=1.0, last_layer_linear=True, last_layer_no_bn=True,
                 aux_loss_func=("ring_loss", "mhe_loss")), B=4, T=30),
    # frame-layer tables whose last backward stage has no weight gradient on the side stream (three layers) / spans two layers (ten)
    dict(kw=dict(loss_func="softmax", frame_layers=((5, 64), (3, 64), (1, 1500))), B=4, T=30),
    dict(kw=dict(loss_func="softmax", frame_layers=((5, 64), (1, 64), (3, 64), (1, 64), (3, 64), (1, 64), (3, 64), (1, 64), (1, 64), (1, 1500))), B=3, T=40),
    dict(kw=dict(loss_func="softmax", network_relu_type="prelu"), B=4, T=30),
    dict(kw=dict(loss_func="softmax", clip_gradient_norm=0.5), B=4, T=30),      # a global-norm clip: the entry point runs the two calls itself
]


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("case", SCHEDULED_CASES, ids=lambda c: "-".join(str(v) for v in list(c["kw"].values())[:2]))
def test_scheduled_update_is_bit_identical_to_backward_then_apply(case, precision):
    """xv_engine_backward_update (what Engine.train_step runs on one GPU: the optimiser step and the next forward's weight copies
    enqueued per backward stage on the side streams) against xv_engine_backward + xv_engine_apply from the same variables, over several
    steps of different batch shapes - the second and later steps start from weight copies the scheduled pass made.  Variables,
    gradients, optimiser state and the following forward's embedding must agree bit for bit (split precision: the entry point falls back
    to the two calls, the equality is then trivial but the path is exercised)."""
    import torch
    from tf_kaldi_speaker_amd import engine as E
    kw = dict(case["kw"])
    N, D = 41, 30
    lf = kw.pop("loss_func")

    def build():
        eng = E.Engine(E.make_config(D, N, loss_func=lf, max_batch=case["B"], max_frames=case["T"] + 8, precision=precision, **kw), device="cuda:0")
        eng.init_variables(seed=3)
        return eng
    a, b = build(), build()
    rs = np.random.RandomState(17)
    lrs = [0.05, 0.02, 0.03, 0.01]
    for i, lr in enumerate(lrs):
        bb = case["B"] - (i % 2)
        tt = case["T"] + 4 * (i % 3)
        x = torch.from_numpy(rs.randn(bb, tt, D).astype(np.float32)).cuda()
        y = torch.from_numpy(rs.randint(0, N, bb).astype(np.int32)).cuda()
        assert a.train_step(x, y, lr, 100 + i) is None                     # scheduled: backward_update
        b.forward(x, True); b.loss(y, 100 + i, True); b.backward(-1); b.apply(lr)      # the two calls
        torch.cuda.synchronize()
        if kw.get("clip_gradient_norm"):
            # the global norm is a float atomicAdd over workgroups (order-dependent in its last bits): two runs of the SAME calls differ there
            same = lambda u, v: torch.allclose(u, v, rtol=2e-6, atol=1e-9)
        else:
            same = torch.equal
        assert same(a.grads, b.grads), "gradients differ at step %d" % i
        assert same(a.variables, b.variables), "variables differ at step %d" % i
        assert same(a.opt_state, b.opt_state), "optimiser state differs at step %d" % i
        assert float(a.raw_loss()) == float(b.raw_loss())
    # an inference forward after a scheduled step uses the copies that step made (plus the first layers', made on demand)
    x = torch.from_numpy(rs.randn(3, case["T"], D).astype(np.float32)).cuda()
    a.forward(x, False); b.forward(x, False)
    emb = "tdnn%d_dense" % ((len(kw["frame_layers"]) if "frame_layers" in kw else 5) + 1)      # the first segment-level layer
    assert (torch.allclose if kw.get("clip_gradient_norm") else torch.equal)(a.endpoint(emb), b.endpoint(emb))
    # a logging step (losses on the pre-update weights) and a step after set_variables go through the plain calls / a full re-copy
    ya = torch.from_numpy(rs.randint(0, N, 3).astype(np.int32)).cuda()
    la = a.train_step(x, ya, 0.01, 200, fetch_losses=True)
    lb = b.train_step(x, ya, 0.01, 200, fetch_losses=True)
    # (the regularisation loss is a float atomicAdd over workgroups: order-dependent in its last bits)
    assert (la[0] == lb[0] or kw.get("clip_gradient_norm")) and abs(la[1] - lb[1]) <= 5e-6 * abs(lb[1])
    a.set_variables({"tdnn/tdnn1_conv/bias": np.full(a.table["tdnn/tdnn1_conv/bias"][0], 0.25, np.float32)})
    b.set_variables({"tdnn/tdnn1_conv/bias": np.full(b.table["tdnn/tdnn1_conv/bias"][0], 0.25, np.float32)})
    a.train_step(x, ya, 0.01, 201)
    b.forward(x, True); b.loss(ya, 201, True); b.backward(-1); b.apply(0.01)
    torch.cuda.synchronize()
    assert (torch.allclose if kw.get("clip_gradient_norm") else torch.equal)(a.variables, b.variables)
    a.close(); b.close()


def test_predict_only_engine_keeps_the_two_slot_arena():
    """an engine without a loss head never runs a backward pass - it must not pay for a dz slot per layer (Trainer.predict_batch
    builds one with 49 152 rows: nine slots of 296 MB against two)."""
    from tf_kaldi_speaker_amd import engine as E
    rows = 49152
    pred = E.Engine(E.make_config(30, 0, max_batch=128, max_frames=4000, max_rows=rows), device="cuda:0")
    train = E.Engine(E.make_config(30, 100, max_batch=128, max_frames=4000, max_rows=rows), device="cuda:0")
    slot = rows * 1504 * 4
    assert train.arena_bytes - pred.arena_bytes > 5 * slot, (train.arena_bytes, pred.arena_bytes)
    assert pred.arena_bytes < 4e9, pred.arena_bytes
    pred.close(); train.close()
