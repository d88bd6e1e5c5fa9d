#!/usr/bin/env python3
"""What a launch that does nothing costs on this GPU (tools/probe: vk_probe_launch_floor): chains of
early-exit kernels in stream order, timed with HIP events (r03: ~3.0 us each, whatever the grid)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from vulcan_amd import api
    lib, s = api.lib(), api.stream()
    pl = C.CDLL(os.path.join(ROOT, "vulcan_amd", "lib", "libvk_probe.so"))
    pl.vk_probe_launch_floor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    pl.vk_probe_launch_floor_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    ctr = torch.zeros(8, dtype=torch.int32, device="cuda")
    sink = torch.zeros(4096 * 256, dtype=torch.float32, device="cuda")

    def ev():
        e = C.c_void_p()
        api.check(lib.vk_event_create(C.byref(e)), "event")
        return e

    for wgs in (1, 72, 256, 1200, 4096):
        for launches in (1, 4, 8):
            replays = 200
            rc = pl.vk_probe_launch_floor(ctr.data_ptr(), sink.data_ptr(), wgs, launches, 20, s)
            assert rc == 0, rc
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            lib.vk_event_record(e0, s)
            pl.vk_probe_launch_floor(ctr.data_ptr(), sink.data_ptr(), wgs, launches, replays, s)
            lib.vk_event_record(e1, s)
            ms = C.c_float()
            lib.vk_event_elapsed_ms(e0, e1, C.byref(ms))
            line = f"workgroups={wgs:5d} chain={launches}: {ms.value * 1e3 / (replays * launches):6.2f} us per launch"
            us = C.c_float()
            rc = pl.vk_probe_launch_floor_graph(ctr.data_ptr(), sink.data_ptr(), wgs, launches, replays, C.byref(us))
            line += f" | captured in a hipGraph: {us.value:6.2f} us per launch" if rc == 0 else f" | hipGraph: HIP error {rc}"
            print(line)


if __name__ == "__main__":
    main()
