"""GPU integration: the whole lptask flow (encoder training with dropout and grad clipping ->
hyperbolic k-means -> diffusion training -> sampling) on a small synthetic graph.  Checks that the
pieces compose and learn; numerics of each piece are pinned in the parity tests."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gcn", [False, True])
def test_lptask_end_to_end(gcn):
    import lptask_synthetic as ex
    r = ex.run(nodes=300, dim=16, epochs=30, diff_epochs=30, timesteps=200, gcn=gcn, log=lambda *_: None)
    assert r["encoder_losses"][-1] < r["encoder_losses"][0]
    assert r["auc_encoder"] > 0.6
    assert r["diffusion_losses"][-1] < r["diffusion_losses"][0]
    assert r["sample_finite"]
    assert sum(r["cluster_sizes"]) == 300
