"""Seeded random configurations, device vs oracle, bit for bit: voxel and truncation
lengths that are not the defaults (block length != truncation), odd table sizes, tilted
poses, depth images with holes and out-of-range values, depth-only / colour / light
integration, raycasts from a second pose."""
import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, frames, make_pair, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


def random_pose(rng, scale=1.0):
    q = rng.normal(size=4) * np.array([0, 0.08, 0.08, 0.08]) * scale + np.array([1, 0, 0, 0])
    q /= np.linalg.norm(q)
    t = rng.normal(size=3) * 0.05 * scale
    return T.Transform.translate(*t) * T.Transform.rotate(*q)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("VK_FUZZ_FIRST", "0")), int(__import__("os").environ.get("VK_FUZZ_LAST", "10"))))
def test_random_configuration(api, orc, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(97, 200)), int(rng.integers(70, 150))
    k = T.Projection.make(0.9 * w + rng.normal() * 5, 0.9 * w + rng.normal() * 5, 0.5 * w + rng.normal() * 3, 0.5 * h + rng.normal() * 3)
    voxel = float(rng.choice([0.004, 0.006, 0.008, 0.0125, 0.02]))
    trunc = float(voxel * rng.choice([2.5, 4.0, 5.0, 7.5]))
    main, excess = int(rng.choice([4099, 8191, 16384, 20011])), int(rng.choice([257, 1021, 4096]))
    y, x = np.mgrid[0:h, 0:w]
    base = 0.8 + 0.6 * rng.random()
    depth = (base + 0.15 * np.sin(x / (7.0 + 10 * rng.random())) * np.cos(y / (5.0 + 10 * rng.random()))
             + 0.002 * rng.normal(size=(h, w))).astype(np.float32)
    depth[rng.random((h, w)) < 0.02] = 0.0                      # holes
    depth[rng.random((h, w)) < 0.01] = 9.0                      # beyond max depth
    depth[: h // 9, : w // 7] = 0.05                            # closer than min depth
    color = rng.random((h, w, 3), dtype=np.float32)
    mode = seed % 3                                             # 0 depth, 1 colour, 2 light
    light = T.Light.make(1.5 + rng.random(), tuple(rng.normal(size=3) * 0.05))
    hv, dv = make_pair(api, orc, main, excess, voxel, trunc)
    tracer = api.Tracer(dv)
    integ = [api.DepthIntegrator, api.ColorIntegrator, api.LightIntegrator][mode](dv)
    if mode == 2:
        integ.light = light
    pose = random_pose(rng)
    for frame_index in range(3):
        pose = random_pose(rng, 0.3) * pose
        hf, df = frames(api, orc, depth, k, pose, color=color)
        hf.compute_normals()
        df.compute_normals()
        for _ in range(2):
            hv.set_view(hf, orc.POLICY_MAXKEY)
            dv.set_view(df)
        orc.integrate_depth(hv, hf)
        if mode == 1:
            orc.integrate_color(hv, hf)
        if mode == 2:
            orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2))
        integ.integrate(df)
        sync()
        assert_volume_equal(dv, hv)
        view = random_pose(rng, 0.2) * pose if frame_index == 1 else pose       # also a view that was not integrated
        hf.depth_to_world = view
        want = orc.trace(hv, hf)
        out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, view)
        tracer.trace(out)
        sync()
        assert np.array_equal(tracer.bounds.cpu().numpy(), want[3])
        assert np.array_equal(out.depth.cpu().numpy(), want[0])
        assert np.array_equal(out.color.cpu().numpy(), want[1])
        assert np.array_equal(out.normals.cpu().numpy(), want[2], equal_nan=True)
    assert dv.visible_count > 50
