#!/usr/bin/env python3
"""vk_probe_rounding (tools/probe): vk_probe.hip's rcp_rn_mid / sqrt_rn_mid against the compiler's 1.0f / x and sqrtf(x) for every
float in [2^-60, 2^60]. Prints the six counters."""
import ctypes as C
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch   # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "vulcan_amd", "lib", os.environ.get("VK_PROBE_LIBRARY", "libvk_probe.so")))
out = torch.zeros(6, dtype=torch.int64, device="cuda")
rc = lib.vk_probe_rounding(C.c_void_p(out.data_ptr()), None)
torch.cuda.synchronize()
v = [int(x) for x in out.cpu()]
print("rc", rc, "reciprocal: tested", v[0], "differ", v[1], "| sqrt: tested", v[2], "differ", v[3])
for name, first in (("reciprocal", v[4]), ("sqrt", v[5])):
    if first:
        bits = first - 1
        print("  first", name, "mismatch at", hex(bits), struct.unpack("f", struct.pack("I", bits))[0])
