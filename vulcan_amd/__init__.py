"""vulcan_amd — MI355X-native (gfx950) implementation of mkaspr/Vulcan's per-frame
fusion + raycast hot path.

The product is the C-ABI shared library declared in include/vk.h
(vulcan_amd/lib/libvk_hip.so, built from vulcan_amd/csrc/*.hip) plus the C++
class layer in vulcan_amd/host/. The Python modules here are thin plumbing used
by tests/ and bench.py: ctypes bindings (api.py) over device memory held in
torch tensors. Importing this package does not load the HIP library; the first
call into vulcan_amd.api does, and raises if it is missing.
"""
__version__ = "0.1.0"
