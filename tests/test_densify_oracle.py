"""oracle/densify_oracle.py against tests/golden/ref_densify.npz = the reference's own GaussianModel.densify_and_prune /
reset_opacity (scene/gaussian_model.py:215-218, 263-413) run on the CPU by tests/golden/make_golden.py part_d."""
import os

import numpy as np
import pytest

from oracle import densify_oracle as do

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_densify.npz")


def load_case(z, tag):
    par = {k: z[f"{tag}_in_{k}"] for k in do.NAMES}
    m = {k: z[f"{tag}_in_m_{k}"] for k in do.NAMES}
    v = {k: z[f"{tag}_in_v_{k}"] for k in do.NAMES}
    mg, mo, ext, pd, big = z[f"{tag}_settings"]
    return par, m, v, z[f"{tag}_in_accum"], z[f"{tag}_in_denom"], z[f"{tag}_noise"], (float(mg), float(mo), float(ext), float(pd), bool(big))


def check_against_golden(z, tag, out, om, ov, stats):
    """rows, order and carried moments exactly; the two computed columns (split xyz / scaling) to float32 rounding"""
    for k in do.NAMES:
        want = z[f"{tag}_out_{k}"]
        assert out[k].shape == want.shape, (k, out[k].shape, want.shape)
        if k in ("xyz", "scaling"):
            np.testing.assert_allclose(out[k], want, rtol=2e-6, atol=2e-6, err_msg=k)
        else:
            assert np.array_equal(out[k], want), k
        assert np.array_equal(om[k], z[f"{tag}_out_m_{k}"]), k
        assert np.array_equal(ov[k], z[f"{tag}_out_v_{k}"]), k
    assert np.array_equal(stats[0], z[f"{tag}_out_accum"]) and np.array_equal(stats[1], z[f"{tag}_out_denom"])
    assert np.array_equal(stats[2], z[f"{tag}_out_max_radii2D"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_reproduces_the_reference(tag):
    z = np.load(GOLD)
    par, m, v, acc, den, noise, (mg, mo, ext, pd, big) = load_case(z, tag)
    out, om, ov, stats, (src, kind, draw, S) = do.densify_and_prune(par, m, v, acc, den, noise, mg, mo, ext, pd, big)
    assert noise.shape[0] == 2 * S
    check_against_golden(z, tag, out, om, ov, stats)
    if tag != "c":
        assert (kind == 1).sum() > 20 and (kind == 2).sum() > 20 and (kind == 0).sum() < len(par["xyz"])
    np.testing.assert_allclose(do.reset_opacity(out["opacity"]), z[f"{tag}_reset_opacity"], rtol=2e-6, atol=2e-6)
    # Adam's step counters survive the surgery untouched (the state dict is re-keyed, not rebuilt)
    assert all(float(z[f"{tag}_out_step_{k}"]) == 3.0 for k in do.NAMES)
