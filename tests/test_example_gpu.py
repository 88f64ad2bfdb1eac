"""Caller integration (as far as the reference's checkout allows): a training loop on the
drop-in Function must actually learn."""
import pytest

pytestmark = pytest.mark.gpu


def test_fit_synthetic_loss_decreases(device):
    from examples.fit_synthetic import fit

    losses = fit(n_gauss=3000, width=96, height=64, depth=20.0, steps=40, seed=3, log=lambda *_: None)
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    assert all(l == l for l in losses)  # no NaN
