"""CPU-side checks of the drop-in boundary: libvk_hip.so loads without a GPU,
exports exactly what include/vk.h declares, the Python bindings cover every
entry point, argument validation works without touching a device, and the
product never reaches into oracle/."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vk.h")


def declared():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"VK_API\s+[\w\s\*]+?\b(vk_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vulcan_amd import api
    lib = api.lib()
    names = declared()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} declared in vk.h but not exported"
    assert sorted(api.EXPORTS) == names, "python bindings out of sync with vk.h"
    out = subprocess.run(["nm", "-D", "--defined-only", api.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T vk_" in l)
    assert exported == names, "library exports symbols vk.h does not declare"


def test_every_entry_point_cites_the_reference():
    text = open(HEADER).read()
    for name in declared():
        if re.match(r"vk_(error_string|version|device_|set_device|stream_|malloc|free|memcpy|memset|event_|probe_|trace_bounds_floats|"
                    r"icp_workspace|trace_compute_block_bounds|test_hooks_|abi_)", name):
            continue
        head = text[:text.index(name + "(")]
        comment = head[head.rindex("/*"):]
        assert "ref:" in comment, f"{name} has no reference citation"


def test_version_and_errors_without_gpu():
    from vulcan_amd import api
    lib = api.lib()
    assert lib.vk_version() == 100
    assert b"invalid argument" in lib.vk_error_string(-1)
    assert lib.vk_volume_initialize(None, None) == -1              # VK_ERR_ARGUMENT, no device touched
    assert lib.vk_integrate_depth(None, None, None, None) == -1
    assert lib.vk_trace_compute_points(None, None, None, 0, 0.0, 0.0, 0.0, None, None, None, None, 0, 0, 0, 0, None) == -1
    # the round-4 entry points validate before they touch a device as well
    assert lib.vk_trace_ahead_requests(None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.vk_volume_set_view_rounds_ahead(None, None, None, 3, None, None) == -1
    assert lib.vk_icp_pyramid_track_frame(None, None, None, None, None, 1, None, None, None, None, None, None, None, None, None) == -1
    assert lib.vk_abi_version() == 7 and lib.vk_abi_check(6, 0, 0, 0) == -2      # VK_ERR_UNSUPPORTED
    # round 5: the cancel of an announced frame, the outcome of the normals' bounded wait
    assert lib.vk_requests_ahead_cancel(None, None, 1, None) == -1 and lib.vk_trace_normals_settle(None, None) == -1
    assert b"bounded wait" in lib.vk_error_string(-6)
    # the launch that times itself takes both events or neither (host state only: nothing is enqueued, no device touched)
    one = C.c_void_p(1)
    assert lib.vk_integrate_time_next(one, None) == -1 and lib.vk_integrate_time_next(None, one) == -1
    assert lib.vk_integrate_time_next(None, None) == 0
    # scratch of a tracer: merged grid + 32 private grids + the normals' row counters (128 lines of 16 words) + the expiry line
    assert lib.vk_trace_bounds_floats(80, 60) == 2 * 4800 * 33 + 128 * 16 + 16
    assert lib.vk_icp_workspace_floats(640, 480) == 4 * 1200 * 32 + 32     # two parities of 1200 slots (256-pixel groups) x 32 {tag, value} words + a pose
    n = C.c_int(-5)
    lib.vk_device_count(C.byref(n))
    assert n.value >= 0


def test_pod_layouts_match_header():
    from vulcan_amd import vk_types as T
    assert C.sizeof(T.Volume) == 8 * 8 + 2 * 4 + 4 * 4
    assert C.sizeof(T.Frame) == 3 * 8 + 4 * 4 + 2 * 16 + 2 * 128 + 8 and T.Frame.content_id.offset == 328
    assert T.Frame.color_width.offset == 32 and T.Frame.depth_projection.offset == 40
    assert C.sizeof(T.Integrator) == 16 and C.sizeof(T.IcpView) == 2 * 8 + 2 * 4 + 16


def test_every_mirrored_struct_has_the_size_the_c_compiler_gives_it(tmp_path):
    """sizeof() of every struct of vk.h that vk_types.py mirrors, asked of gcc, against ctypes: a field added on one side only
    (vk_view_bounds and vk_requests_ahead both grew in round 4) fails here, on the CPU, not as a corrupted launch on the GPU."""
    from vulcan_amd import vk_types as T
    pairs = [("vk_projection", T.Projection), ("vk_transform", T.Transform), ("vk_light", T.Light), ("vk_volume", T.Volume),
             ("vk_frame", T.Frame), ("vk_integrator", T.Integrator), ("vk_view_bounds", T.ViewBounds), ("vk_color_view", T.ColorView),
             ("vk_light_terms", T.LightTerms), ("vk_track_poll", T.TrackPoll), ("vk_light_prep", T.LightPrep),
             ("vk_rig_exchange", T.RigExchange), ("vk_requests_ahead", T.RequestsAhead), ("vk_test_hooks", T.TestHooks),
             ("vk_color_pose", T.ColorPose), ("vk_pyramid_ahead", T.PyramidAhead), ("vk_detector", T.Detector), ("vk_detect_state", T.DetectState), ("vk_icp_view", T.IcpView)]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "vk.h"\nint main(void) {\n' +
                   "".join(f'  printf("{name} %zu\\n", sizeof({name}));\n' for name, _ in pairs) +
                   f'  printf("abi %d ctr %d\\n", VK_ABI_VERSION, VK_CTR_COUNT);\n  return 0;\n}}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split(None, 1) for line in subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines())
    for name, mirror in pairs:
        assert int(out[name]) == C.sizeof(mirror), (name, out[name], C.sizeof(mirror))
    abi, _, ctr = out["abi"].split()
    assert int(abi) == T.VK_ABI_VERSION and int(ctr) == T.VK_CTR_COUNT


def test_product_does_not_use_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "vulcan_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"import\s+oracle|from\s+oracle|from\s+\.+\s*oracle|oracle[/.]\w|liboracle|\borc_\w+\s*\(|"
                             r"#\s*include[^\n]*oracle", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    out = subprocess.run(["ldd", os.path.join(ROOT, "vulcan_amd", "lib", "libvk_hip.so")],
                         stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from vulcan_amd import api
    monkeypatch.setattr(api, "_LIB", None)
    monkeypatch.setattr(api, "LIB_PATH", str(tmp_path / "libvk_hip.so"))
    try:
        api.lib()
    except api.VkError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected VkError")


# ---- libvk_comm.so (include/vk_comm.h): the rig's all-reduce ------------------------

def test_comm_library_exports_and_loopback():
    """Loads without RCCL or a GPU, exports exactly what vk_comm.h declares, validates
    arguments, and a single-rank communicator is a loopback (the sum over one rank)."""
    import numpy as np
    from vulcan_amd import comm
    text = open(os.path.join(ROOT, "include", "vk_comm.h")).read()
    names = sorted(set(re.findall(r"VK_API\s+[\w\s\*]+?\b(vk_comm_\w+)\s*\(", text)))
    assert names == sorted(comm.EXPORTS) and len(names) == 13
    out = subprocess.run(["nm", "-D", "--defined-only", comm.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert sorted(l.split()[-1] for l in out.splitlines() if " T vk_" in l) == names
    assert "rccl" not in subprocess.run(["ldd", comm.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout   # bound at run time
    lib = comm.lib()
    h = C.c_void_p()
    assert lib.vk_comm_init(C.byref(h), None, 0, 0) == -1          # world < 1
    assert lib.vk_comm_init(C.byref(h), None, 2, 2) == -1          # rank out of range
    assert lib.vk_comm_init(C.byref(h), None, 0, 2) == -1          # world > 1 needs an id
    assert lib.vk_comm_allreduce_system(None, None, 48, None) == -1
    assert lib.vk_comm_exchange_attach(None, None) == -1 and lib.vk_comm_exchange_detach(None, None) == -1
    assert lib.vk_comm_exchange_create(None, 0, 1, None) == -1 and lib.vk_comm_exchange_attach_handles(None, None) == -1
    # 1, 2, ... 2^22 - 2, 1: never 0, and the parity alternates across the wrap (vk_rig_protocol.h)
    last = (1 << 22) - 2
    assert lib.vk_comm_exchange_next_sequence(1) == 2 and lib.vk_comm_exchange_next_sequence(last - 1) == last
    assert lib.vk_comm_exchange_next_sequence(last) == 1 and last % 2 == 0
    assert b"invalid argument" in lib.vk_comm_error_string(-1)
    c = comm.Communicator(None, 0, 1)
    r, w = C.c_int(-1), C.c_int(-1)
    assert lib.vk_comm_rank(c.handle, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
    assert c.rccl_count() == 1
    buf = np.arange(48, dtype=np.float32)
    assert lib.vk_comm_reduce_hook(buf.ctypes.data_as(C.c_void_p), 48, c.handle, None) == 0
    assert np.array_equal(buf, np.arange(48, dtype=np.float32))
    c.close()
