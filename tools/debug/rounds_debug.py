import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from vulcan_amd import api, vk_types as T
from oracle import oracle as orc
orc.build(); orc.lib(); api.lib()
K = T.Projection.make(272.0, 272.0, 155.6, 117.4)
w, h = 320, 240
depth = scenes.sphere(2 * w, 2 * h)[::2, ::2].copy()
pose = scenes.tracer_test_pose()
hf = orc.HostFrame(depth, K, pose); df = api.Frame(depth, K, pose)
def blocks(entries):
    idx = np.nonzero(entries["data"] >= 0)[0]
    return {tuple(int(c) for c in entries["block"]["origin"][i]): int(i) for i in idx}
hv = orc.HostVolume(1024, 16384, voxel_length=0.01, truncation_length=0.04)
per_round = []
prev = {}
for r in range(3):
    hv.set_view(hf, orc.POLICY_MAXKEY)
    cur = blocks(hv.hash_entries)
    new = {b: i for b, i in cur.items() if b not in prev}
    per_round.append(new)
    print("oracle round", r + 1, "requests", hv.counters[T.VK_CTR_REQUESTS], "new", len(new), "total", len(cur))
    prev = cur
dv = api.Volume(1024, 16384, voxel_length=0.01, truncation_length=0.04)
dv.set_view(df, rounds=3)
torch.cuda.synchronize()
print("device counters", dv.read_counters())
dcur = blocks(dv.host_entries())
missing = [b for b in prev if b not in dcur]
extra = [b for b in dcur if b not in prev]
print("missing on device", missing, "extra", extra)
K1 = 1024
def hsh(b): return (((b[0] * 73856093) & 0xffffffff) ^ ((b[1] * 19349669) & 0xffffffff) ^ ((b[2] * 83492791) & 0xffffffff)) % K1
for b in missing:
    hb = hsh(b)
    print("block", b, "bucket", hb, "allocated by the oracle in round", [i + 1 for i, n in enumerate(per_round) if b in n])
    for i, n in enumerate(per_round):
        same = [x for x in n if hsh(x) == hb]
        print("   oracle round", i + 1, "allocated in this bucket:", same)
    same_dev = [x for x in dcur if hsh(x) == hb]
    print("   device holds in this bucket:", same_dev)
