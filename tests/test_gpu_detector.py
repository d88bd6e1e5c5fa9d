"""vulcan::Detector on the device (vk_detect*) vs the oracle, bit for bit: counts,
surviving points in input order, centre, limit and position."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import api, sync  # noqa: F401  (fixture)
from test_oracle_detector import cloud
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def run_both(api, orc, pts, params, filter_only=False):
    import torch
    det = api.Detector()
    C.memmove(C.byref(det.params), C.byref(params), C.sizeof(T.Detector))
    dev_pts = torch.from_numpy(np.ascontiguousarray(pts, np.float32).reshape(-1, 3)).cuda()
    if filter_only:
        det.filter(dev_pts)
    else:
        det.enqueue(dev_pts)
    sync()
    got = det.read_state()
    want, want_inliers = orc.detect(pts, params)
    assert (got.filtered_count, got.inlier_count) == (want.filtered_count, want.inlier_count)
    got_inliers = det.inliers[: got.inlier_count].cpu().numpy()
    assert np.array_equal(bits(got_inliers), bits(want_inliers))
    assert np.array_equal(bits(got.center), bits(want.center))
    assert np.array_equal(bits([got.limit, got.squared_error]), bits([want.limit, want.squared_error]))
    if not filter_only:
        assert got.detected == want.detected
        assert np.array_equal(bits(got.position), bits(want.position)) or \
            (np.isnan(np.array(got.position)).all() and np.isnan(np.array(want.position)).all())
    return got


@pytest.mark.parametrize("n", [1, 63, 255, 256, 4095, 4096, 4097, 50000, 307200])
def test_detect_matches_oracle(api, orc, n):
    d = T.Detector.default()
    d.min_inlier_count = 10
    pts = cloud(n, 100 + n, centre=(0.3, -0.2, 1.0), sigma=0.3)
    pts[::17] += 3.0                    # outside the 2 m radius
    got = run_both(api, orc, pts, d)
    assert got.filtered_count < n or n < 17


def test_detect_empty_and_all_rejected(api, orc):
    d = T.Detector.default()
    run_both(api, orc, np.zeros((0, 3), np.float32), d)
    got = run_both(api, orc, np.full((5000, 3), 9.0, np.float32), d)
    assert got.filtered_count == 0 and got.detected == 0


def test_detect_intervals_and_quirk(api, orc):
    pts = cloud(20000, 7, centre=(0.3, 1.0, 1.5), sigma=0.2)
    d = T.Detector.default()
    d.radius = 0.0
    d.bounds[0][0], d.bounds[0][1] = 0.1, 0.5
    d.bounds[1][0], d.bounds[1][1] = 0.2, 0.45      # upstream: tested against x as well
    run_both(api, orc, pts, d)
    d.bounds_use_own_axis = 1
    d.bounds[1][0], d.bounds[1][1] = 0.8, 1.2
    d.bounds[2][0], d.bounds[2][1] = 1.2, 1.9
    got = run_both(api, orc, pts, d)
    assert 0 < got.inlier_count < 20000


def test_filter_stage_alone(api, orc):
    run_both(api, orc, cloud(30000, 9, sigma=0.5), T.Detector.default(), filter_only=True)


def test_detector_reuse_with_smaller_cloud(api, orc):
    """A second call on the same object with fewer points must not see the first call's scratch."""
    import torch
    d = T.Detector.default()
    det = api.Detector()
    a = cloud(100000, 11, sigma=0.4)
    b = cloud(3000, 12, sigma=0.1)
    for pts in (a, b, a):
        pos = det.detect(torch.from_numpy(pts).cuda())
        want, _ = orc.detect(pts, d)
        assert np.array_equal(bits(pos), bits(want.position))
