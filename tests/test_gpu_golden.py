"""The device path against the COMMITTED golden vectors (tests/golden/scenes_160x120.npz):
SetView x5, depth + colour integration, raycast, normals, mesh extraction and ICP residuals
on the four synthetic scenes of SURVEY.md §8(d), bit for bit — without building, loading or
calling the oracle (parity does not hinge on liboracle.so being rebuilt on the GPU box)."""
import os
import sys

import numpy as np
import pytest

from test_gpu_parity import api, sync  # noqa: F401
from vulcan_amd import vk_types as T

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_fixtures as mf  # noqa: E402
from test_golden import compare  # noqa: E402

pytestmark = pytest.mark.gpu


def device_backend(api):
    import torch

    def run(depth, color, k, pose):
        dv = api.Volume(mf.MAIN, mf.EXCESS, voxel_length=mf.VOXEL, truncation_length=mf.TRUNC)
        df = api.Frame(depth, k, pose, color=color)
        df.compute_normals()
        for _ in range(5):
            dv.set_view(df)
        integ = api.ColorIntegrator(dv)
        tracer = api.Tracer(dv)
        integ.integrate(df)
        out = api.Frame(torch.zeros((mf.H, mf.W), dtype=torch.float32, device="cuda"), k, pose)
        tracer.trace(out)
        ex = api.Extractor(dv)
        ex.all_allocated = True
        mesh = ex.extract()
        tracker = api.DepthTracker()
        tracker.keyframe = out
        moved = api.Frame(depth, k, T.Transform.translate(0.002, -0.001, 0.001) * pose, normals=df.normals)
        residuals = tracker.compute_residuals(moved)
        sync()
        points, faces = mesh.host()
        return dict(frame_normals=df.normals.cpu().numpy(), depth=out.depth.cpu().numpy(), color=out.color.cpu().numpy(),
                    normals=out.normals.cpu().numpy(), bounds=tracer.bounds.cpu().numpy(), counters=dv.read_counters()[:8],
                    visible=np.sort(dv.visible()), voxels_sha256=mf.digest(dv.host_voxels()),
                    entries_sha256=mf.digest(dv.host_entries()), visibility_sha256=mf.digest(dv.host_visibility()),
                    mesh_points_sha256=mf.digest(points), mesh_faces_sha256=mf.digest(faces),
                    mesh_counts=np.array([len(points), len(faces), ex.skipped], dtype=np.int32),
                    icp_residuals=residuals.cpu().numpy())
    return run


@pytest.mark.parametrize("name", mf.SCENES)
def test_device_reproduces_the_golden_vectors(api, name):
    golden = np.load(mf.FILE)
    got = mf.run_scene(device_backend(api), name)
    # the pending-update counters are internal to the device path (VK_CTR_PENDING_*)
    got["counters"] = got["counters"].copy()
    want = golden[f"{name}/counters"].copy()
    for i in (T.VK_CTR_PENDING_ALL, T.VK_CTR_PENDING_EXCESS, T.VK_CTR_REQUESTS, T.VK_CTR_PATCHES):
        got["counters"][i] = want[i]
    compare(got, golden, name)


def test_device_reproduces_the_step_cap_vectors(api):
    """The march's 500-step cap (tracer.cu:437-442) against tests/golden/step_cap_64x48.npz, without the oracle: the
    device's own SetView allocates the slab (its table must have the golden digest), the voxels are the closed form of
    tests/scenes.py, the raycast must paint the capped half (1, 0, 0) at depth 0 and hit the surface in the other."""
    import torch
    import step_cap
    golden = np.load(mf.STEP_CAP_FILE)
    dv, df = step_cap.build_device(api)
    tracer = api.Tracer(dv)
    w, h = df.width, df.height
    out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), df.depth_projection, df.depth_to_world)
    tracer.trace(out)
    sync()
    got = dict(depth=out.depth.cpu().numpy(), color=out.color.cpu().numpy(), normals=out.normals.cpu().numpy(),
               bounds=tracer.bounds.cpu().numpy(), visible=np.sort(dv.visible()),
               entries_sha256=mf.digest(dv.host_entries()), voxels_sha256=mf.digest(dv.host_voxels()))
    compare(got, golden, "step_cap")
    capped, hit = step_cap.classify(got["depth"], got["color"])
    assert capped.sum() > 1200 and hit.sum() > 1200


def test_debug_build_reproduces_the_golden_vectors():
    """libvk_hip_debug.so (VK_DEBUG_SYNC: device synchronisation + error check after every
    launch, ref: device.h:48-52) is the same ABI and must give the same bits. Run in a child
    process, because a process binds one build of the library."""
    import subprocess
    root = os.path.dirname(HERE)
    lib = os.path.join(root, "vulcan_amd", "lib", "libvk_hip_debug.so")
    assert os.path.exists(lib), "run __graft_entry__.build() first"
    env = dict(os.environ, VK_HIP_LIBRARY=lib)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); "
            "from vulcan_amd import api; import test_gpu_golden as t, numpy as np; "
            "assert api.lib() and api.LIB_PATH.endswith('libvk_hip_debug.so'); "
            "[t.test_device_reproduces_the_golden_vectors(api, n) for n in ('plane', 'ripple')]; print('debug build ok')"
            % (root, HERE))
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "debug build ok" in p.stdout, p.stdout[-3000:]
