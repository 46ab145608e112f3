"""GPU: the HIP engine against the fixtures made by executing the reference's graph-building code on the TensorFlow stand-in
(tests/golden/make_graph_golden.py; see tests/test_graph_golden.py for what those fixtures are and are not).  Variables are
loaded under their TF names through the checkpoint name mapping, the eight arrays are fed, and the engine's eval-mode and
training-mode outputs, loss, summaries and EMA updates are compared with what the reference's code computed in float32."""
import numpy as np
import pytest

import alignnet3d
from alignnet3d import tf_bundle as tb
from alignnet3d.engine import SUMMARY_NAMES
from tests.helpers import compare_forward
from tests.test_graph_golden import G, META, CASES, LABELS, US, EP, case_cfg, tf_name

pytestmark = pytest.mark.gpu


def _engine(case):
    cfg = case_cfg(case)
    eng = alignnet3d.Engine(cfg)
    mapping, missing = tb.map_variables(eng.variables(), [v["name"] for v in META[case]["variables"]])
    assert not missing and len(mapping) == len(META[case]["variables"])      # every variable the reference's code created, no other
    for name, (r, c), _ in eng.variables():
        eng.set_variable(name, G["%s/f32/var/%s" % (case, mapping[name])].reshape(r, c))
    d = {k: G["%s/f32/in/%s" % (case, k)] for k in ("pcs1", "pcs2") + LABELS}
    return eng, d


@pytest.mark.parametrize("case", CASES)
def test_eval_forward_and_loss_match_executed_reference_graph(gpu_required, case):
    eng, d = _engine(case)
    ep = eng.forward(d["pcs1"], d["pcs2"])
    ref = {k: G["%s/f32/eval/ep/%s" % (case, k)] for k in EP}
    worst, unstable = compare_forward(ep, ref, META[case]["num_bins"])     # bar 1e-4 (north_star)
    loss, summ = eng.eval_loss(d, META[case]["B"])
    rl = float(G["%s/f32/eval/loss" % case])
    assert SUMMARY_NAMES == tuple(META[case]["summary_tags"])
    if unstable == 0:
        assert abs(loss - rl) <= 2e-4 * max(1.0, abs(rl)), (loss, rl)
        for tag, r in zip(META[case]["summary_tags"], G["%s/f32/eval/summaries" % case]):
            assert abs(summ[tag] - r) <= 5e-4 * max(1.0, abs(r)), (tag, summ[tag], r)
    print(case, "eval: worst abs err", max(worst.values()), "unstable", unstable, "loss", loss, rl)
    eng.close()


@pytest.mark.parametrize("case", CASES)   # incl. the four- / five-layer case: the general-depth training path
def test_train_forward_loss_ema_match_executed_reference_graph(gpu_required, case):
    eng, d = _engine(case)
    assert abs(eng.state()["bn_decay"] - META[case]["bn_decay"]) < 1e-7
    u = [G["%s/f32/dropout_u/%d" % (case, i)] for i in range(len(US))]
    res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, u)
    for k in EP:
        np.testing.assert_allclose(res[k], G["%s/f32/train/ep/%s" % (case, k)], rtol=3e-4, atol=3e-4, err_msg=k)
    rl = float(G["%s/f32/train/loss" % case])
    assert abs(res["loss"] - rl) <= 2e-4 * max(1.0, abs(rl)), (res["loss"], rl)
    for tag, r in zip(META[case]["summary_tags"], G["%s/f32/train/summaries" % case]):
        assert abs(res["summaries"][tag] - r) <= 5e-4 * max(1.0, abs(r)), (tag, res["summaries"][tag], r)
    worst = 0.0
    for name, (r, c), trainable in eng.variables():
        if trainable:
            continue
        ref = G["%s/f32/train/ema_after/%s" % (case, tf_name(name))]
        got = eng.get_variable(name)
        np.testing.assert_allclose(got.ravel(), ref.ravel(), rtol=2e-4, atol=2e-5, err_msg=name)
        worst = max(worst, float(np.abs(got.ravel() - ref.ravel()).max()))
    print(case, "train: loss", res["loss"], rl, "worst EMA err", worst)
    eng.close()
