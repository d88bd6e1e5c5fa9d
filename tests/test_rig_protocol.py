"""The rig's exchange inside the one-launch Gauss-Newton loop (vk.h vk_rig_exchange,
vulcan_amd/csrc/vk_rig_protocol.h): protocol rehearsal on host threads — slot indexing, tags, and
the buffering by step parity and Track parity — and the layout the three places that spell it out
must agree on. More than one rank has never run on hardware (this pool hands out single-GPU boxes);
tests/test_gpu_rig_exchange.py runs the kernel side with a rig of one."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vulcan_amd", "host", "bin", "rig_rehearsal")


@pytest.mark.parametrize("world,tracks,steps", [(2, 1500, 1), (2, 40, 200), (3, 300, 3), (4, 40, 20), (8, 10, 7), (1, 5, 9)])
def test_protocol_rehearsal(world, tracks, steps):
    """`world` host threads publish and gather the 27 words for `tracks` Tracks of `steps` steps with
    random pauses: every gathered total must be the rank-ordered sum of exactly that step's values.
    One step per Track is the case that deadlocks without the Track-parity buffers."""
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    proc = subprocess.run([EXE, str(world), str(tracks), str(steps), "5"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=240)
    assert proc.returncode == 0 and " 0 failure(s)" in proc.stdout, proc.stdout[-2000:]


@pytest.mark.parametrize("world,tracks,steps,start,abort_every", [
    (2, 600, 1, (1 << 22) - 300, 0),        # the wrap 2^22 - 2 -> 1 with one step per Track: the parities must alternate across it
    (4, 60, 6, (1 << 22) - 20, 0),
    (2, 120, 8, 1, 5),                      # every fifth Track aborted half way: the Tracks after it are intact
    (3, 90, 4, (1 << 22) - 40, 4),          # aborts and the wrap together
    (8, 24, 3, 7, 3)])
def test_sequence_wrap_and_aborted_tracks(world, tracks, steps, start, abort_every):
    """the sequence number advances after EVERY Track, aborted ones included (a retry under the old number would
    accept the aborted attempt's words), and wraps 2^22 - 2 -> 1 so that the parity that picks the buffers alternates."""
    proc = subprocess.run([EXE, str(world), str(tracks), str(steps), "3", str(start), str(abort_every)], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=240)
    assert proc.returncode == 0 and " 0 failure(s)" in proc.stdout, proc.stdout[-2000:]
    if abort_every:
        assert f" {tracks // abort_every} aborted" in proc.stdout, proc.stdout


def test_layout_is_spelled_the_same_everywhere():
    from vulcan_amd import vk_types as T
    header = open(os.path.join(ROOT, "vulcan_amd", "csrc", "vk_rig_protocol.h")).read()
    ranks = int(re.search(r"#define VK_RIG_MAX_RANKS (\d+)", header).group(1))
    words = int(re.search(r"#define VK_RIG_WORDS (\d+)", header).group(1))
    vk = open(os.path.join(ROOT, "include", "vk.h")).read()
    assert int(re.search(r"#define VK_RIG_MAX_RANKS (\d+)", vk).group(1)) == ranks
    comm = open(os.path.join(ROOT, "vulcan_amd", "csrc", "vk_comm.cpp")).read()
    m = re.search(r"kRigMaxRanks = (\d+), kRigAreaBytes = ([\d \*]+)", comm)
    assert int(m.group(1)) == ranks and eval(m.group(2)) == 4 * ranks * words * 8
    assert C.sizeof(T.RigExchange) == 8 * ranks + 4 + 4 + 4 + 4 and T.RigExchange.rank.offset == 8 * ranks
    from vulcan_amd import api
    assert api.lib().vk_rig_area_bytes() == 4 * ranks * words * 8
