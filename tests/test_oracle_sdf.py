"""Pins the oracle's SDF builder on the reference's own golden vector
(reference tests/sdf/sdf_test.cpp:6-33: testdata.nrrd, TF `value > 800`, values.x)."""
import numpy as np

from cl_volume_renderer_amd import scene


def test_sdf_matches_reference_golden_vector(orc, sdf_golden):
    vol, gold = sdf_golden
    assert vol.shape == (38, 35, 38) and gold.size == 50540  # sdf_test.cpp:6
    tf = orc.parse_tf(scene.TF_TEST_VALUE_GT_800)
    sdf, n_launches, counts = orc.sdf_build(vol, tf)
    assert np.array_equal(sdf.reshape(-1).astype(np.int32), gold)
    # convergence trace recorded in SURVEY.md Appendix B.6
    assert n_launches == 13
    assert counts.tolist() == [21970, 18741, 14640, 10664, 7357, 5120, 3759, 2909, 2143, 1366, 671, 158, 0]
    assert sdf.min() == -8 and sdf.max() == 12


def test_sdf_max_iterations_and_sign(orc):
    vol = scene.phantom(32)
    tf = orc.parse_tf(scene.tf_default_source())
    sdf, n, _ = orc.sdf_build(vol, tf)
    event = (vol >= 500) & (vol <= 1200)
    assert np.all((sdf < 0) == event)          # sign encodes inside / outside of the event set
    assert np.abs(sdf).max() <= 16             # max_iterations = min(max(dims)/2, 127)
    assert np.abs(sdf).min() >= 1
    assert n % 2 == 1                          # the host loop only stops on an odd layer


def test_sdf_gradient_tf_differs_from_value_tf(orc):
    vol = scene.phantom(24)
    a, _, _ = orc.sdf_build(vol, orc.parse_tf(scene.tf_default_source()))
    b, _, _ = orc.sdf_build(vol, orc.parse_tf(scene.tf_gradient_source()))
    assert not np.array_equal(a, b)
