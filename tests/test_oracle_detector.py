"""Oracle restatement of vulcan::Detector (src/detector.cu) against hand-derived
answers. PARITY UNPINNED: the reference has no Detector test or vector
(tests/detector_test.cu is empty), so these cases are derived from the source
text: FilterKernel :14-35, GetValidPosition :131-141, Filter :152-188."""
import numpy as np
import pytest

from vulcan_amd import vk_types as T


def cloud(n, seed, centre=(0.3, -0.2, 1.0), sigma=0.05):
    rng = np.random.default_rng(seed)
    return (rng.normal(0.0, sigma, size=(n, 3)) + np.asarray(centre)).astype(np.float32)


def test_defaults_match_reference_constructor():
    d = T.Detector.default()                      # detector.cu:66-72, 214-221
    assert d.radius == 2.0 and d.min_inlier_count == 100
    assert tuple(d.origin) == (0.0, 0.0, 0.0)
    assert all(tuple(d.bounds[a]) == (1.0, -1.0) for a in range(3))
    assert d.bounds_use_own_axis == 0


def test_too_few_points_is_nan(orc):
    state, inliers = orc.detect(cloud(50, 0), T.Detector.default())
    assert state.filtered_count == 50 and state.detected == 0
    assert np.isnan(np.array(state.position)).all()      # detector.cu:143-147


def test_empty_cloud(orc):
    state, inliers = orc.detect(np.zeros((0, 3), np.float32), T.Detector.default())
    assert (state.filtered_count, state.inlier_count, state.detected) == (0, 0, 0)
    assert np.isnan(np.array(state.position)).all()


def test_position_is_mean_of_absolute_values(orc):
    """GetValidPosition uses cublasSasum = sum |x| (detector.cu:137-139): a cloud
    centred at negative y reports a positive y."""
    pts = cloud(5000, 1, centre=(0.3, -0.2, 1.0), sigma=0.01)
    state, inliers = orc.detect(pts, T.Detector.default())
    assert state.detected == 1
    expect = np.abs(inliers.astype(np.float64)).mean(axis=0)
    np.testing.assert_allclose(np.array(state.position), expect, rtol=2e-6)
    assert state.position[1] > 0


def test_radius_filter_and_sigma_removal(orc):
    pts = np.concatenate([cloud(4000, 2, centre=(0.5, 0.5, 0.5), sigma=0.02),
                          np.full((10, 3), 5.0, np.float32)])          # 10 points outside the 2 m radius
    d = T.Detector.default()
    state, inliers = orc.detect(pts, d)
    assert state.filtered_count == 4000
    kept = pts[:4000]
    centre = np.abs(kept.astype(np.float64)).sum(axis=0) / 4000
    dist = np.linalg.norm(kept.astype(np.float64) - centre, axis=1)
    limit = 1.5 * np.sqrt((dist ** 2).sum() / 4000)                   # detector.cu:177-179
    np.testing.assert_allclose(np.array(state.center), centre, rtol=2e-6)
    np.testing.assert_allclose(state.limit, limit, rtol=1e-5)
    margin = np.abs(dist - limit) > 1e-5
    expect = dist <= limit
    got = np.zeros(4000, bool)
    # survivors keep input order
    j = 0
    for i in range(4000):
        if j < len(inliers) and np.array_equal(kept[i], inliers[j]):
            got[i] = True
            j += 1
    assert j == len(inliers) == state.inlier_count
    assert (got == expect)[margin].all()


def test_interval_quirk_all_axes_test_x(orc):
    """detector.cu:26-28 tests point[0] against the y and z intervals too."""
    pts = cloud(2000, 3, centre=(0.3, 1.0, 1.5), sigma=0.01)
    d = T.Detector.default()
    d.bounds[1][0], d.bounds[1][1] = 0.9, 1.1                     # y interval around the cloud's y
    state, _ = orc.detect(pts, d)
    assert state.filtered_count == 0                                # x ~ 0.3 is outside [0.9, 1.1]
    d.bounds_use_own_axis = 1
    state, _ = orc.detect(pts, d)
    assert state.filtered_count == 2000


def test_fixed_tree_sum_is_close_to_exact(orc):
    pts = cloud(300000, 4, sigma=0.2)
    d = T.Detector.default()
    d.radius = 0.0                                                  # <= 0: no radius test (detector.cu:25)
    state, _ = orc.detect(pts, d)
    assert state.filtered_count == 300000
    exact = np.abs(pts.astype(np.float64)).sum(axis=0) / 300000
    np.testing.assert_allclose(np.array(state.center), exact, rtol=1e-6)
