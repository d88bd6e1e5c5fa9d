"""Long runs of Integrate with the weight caps the shipped app sets (apps/vulcan/vulcan.cu:92-93:
SetMaxDistanceWeight(100), SetMaxColorWeight(16)) and with caps beyond the integrate kernel's
reciprocal table (vk_integrate.hip kReciprocals: weights below 128 divide through a table of
RN64(1 / n), a wave that meets a larger weight takes the plain divisions of
depth_integrator.cu:69-75 / color_integrator.cu:118-131). Every voxel byte against the oracle at
the integration counts around both switches: 31 / 32 / 33 (the table's size until round 3),
127 / 128 / 129 / 130 (its size now) and the end of the run.
"""
import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, frames, make_pair, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu

LIGHT = (2.0, (0.025, 0.08, 0.0))          # apps/vulcan/vulcan.cu:87-88
W, H = 160, 120


def two_frames(api, orc):
    """Two views of a rippled wall from the same pose, 6 mm apart in depth: the running averages
    move at every integration, the visible set does not."""
    k = T.Projection.make(136.0, 136.0, 80.0, 60.0)
    y, x = np.mgrid[0:H, 0:W]
    base = (1.5 + 0.05 * np.cos(x / 17.0) * np.sin(y / 13.0)).astype(np.float32)
    pairs = []
    for shift, lo in ((0.0, 0.2), (0.006, 0.35)):
        depth = (base + np.float32(shift)).astype(np.float32)
        color = scenes.checker_color(W, H, lo, 0.9)
        hf, df = frames(api, orc, depth, k, T.Transform.identity(), color=color)
        hf.compute_normals()
        df.compute_normals()
        pairs.append((hf, df))
    return pairs


def run(api, orc, kind, caps, count, checkpoints):
    pairs = two_frames(api, orc)
    hv, dv = make_pair(api, orc, 4096, 1024, 0.008, 0.04)
    for _ in range(4):
        for hf, df in pairs:
            hv.set_view(hf, orc.POLICY_MAXKEY)
            dv.set_view(df)
    assert_volume_equal(dv, hv)
    params = T.Integrator(0.1, 5.0, float(caps[0]), float(caps[1]))
    light = T.Light.make(*LIGHT)
    integ = {"depth": api.DepthIntegrator, "color": api.ColorIntegrator, "light": api.LightIntegrator}[kind](dv)
    integ.params = params
    if kind == "light":
        integ.light = light
    masks = [orc.light_frame_mask(hf, 0.2) for hf, _ in pairs] if kind == "light" else None
    orc.set_threads(16)
    for n in range(1, count + 1):
        hf, df = pairs[n & 1]
        orc.integrate_depth(hv, hf, params)
        if kind == "color":
            orc.integrate_color(hv, hf, params)
        elif kind == "light":
            orc.integrate_light_color(hv, hf, light, masks[n & 1], params)
        integ.integrate(df)
        if n in checkpoints or n == count:
            sync()
            assert dv.host_voxels().tobytes() == hv.voxels.tobytes(), f"{kind}, caps {caps}: voxels differ after {n} integrations"
    orc.set_threads(1)
    return hv


@pytest.mark.parametrize("kind", ["depth", "color", "light"])
def test_the_apps_weight_caps_over_forty_frames(api, orc, kind):
    """vulcan.cu:92-93: caps 100 / 16, 45 integrations — past weight 32, where the round-3 kernel left
    its reciprocal table."""
    hv = run(api, orc, kind, (100, 16), 45, {1, 16, 17, 31, 32, 33, 34, 40})
    assert hv.voxels["distance_weight"].max() == 45
    if kind != "depth":
        assert hv.voxels["color_weight"].max() == 16
        assert (hv.voxels["color_weight"] == 16).sum() > 10000


@pytest.mark.parametrize("kind", ["depth", "color", "light"])
def test_weights_beyond_the_reciprocal_table(api, orc, kind):
    """Caps 200 / 200 (the values the app's commented-out lines hold, vulcan.cu:81-82), 135 integrations:
    from the 129th on every band voxel divides by more than the table holds and the waves take the
    plain divisions, for the distance and for the colour."""
    hv = run(api, orc, kind, (200, 200), 135, {31, 32, 33, 126, 127, 128, 129, 130, 131})
    assert hv.voxels["distance_weight"].max() == 135
    if kind != "depth":
        assert hv.voxels["color_weight"].max() == 135
        assert (hv.voxels["color_weight"] >= 129).sum() > 10000


def test_a_saturated_cap_above_the_table_keeps_the_plain_divisions(api, orc):
    """Cap 130: the weights stop at 130 and every further integration divides by 131 — the steady state of a
    long run whose cap is past the table."""
    hv = run(api, orc, "light", (130, 130), 140, {129, 130, 131, 132})
    assert hv.voxels["distance_weight"].max() == 130 and hv.voxels["color_weight"].max() == 130
