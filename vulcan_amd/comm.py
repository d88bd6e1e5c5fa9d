"""ctypes binding of libvk_comm.so (include/vk_comm.h): the one collective of the hot path,
the all-reduce of the packed ICP normal system of a rigid multi-camera rig, issued from C
on the compute stream — no Python between Gauss-Newton iterations.

    comm = Communicator.from_torch_group(rank, world)     # or Communicator(id_bytes, rank, world)
    tracker.comm = comm                                   # api.DepthTracker / ColorTracker / LightTracker
    tracker.track(frame)

A host without torch.distributed distributes the 128-byte id itself (file, MPI, socket).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvk_comm.so")
ID_BYTES = 128
IPC_HANDLE_BYTES = 64
EXPORTS = ("vk_comm_unique_id", "vk_comm_init", "vk_comm_rank", "vk_comm_allreduce_system",
           "vk_comm_reduce_hook", "vk_comm_count", "vk_comm_exchange_attach", "vk_comm_exchange_detach", "vk_comm_destroy",
           "vk_comm_error_string", "vk_comm_exchange_create", "vk_comm_exchange_attach_handles",
           "vk_comm_exchange_next_sequence")
_LIB = None


class CommError(RuntimeError):
    pass


def agree_over_torch_group(group=None):
    """An `agree` for Communicator (see there): logical OR of a flag over a torch.distributed group."""
    import torch
    import torch.distributed as dist

    def agree(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return bool(int(t.item()))
    return agree


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CommError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        if "VK_RCCL_LIBRARY" not in os.environ:
            # share the RCCL PyTorch already carries instead of loading a second copy
            try:
                import torch
                bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
                if os.path.exists(bundled):
                    os.environ["VK_RCCL_LIBRARY"] = bundled
            except ImportError:
                pass
        h = C.CDLL(LIB_PATH)
        P, I = C.c_void_p, C.c_int
        h.vk_comm_unique_id.argtypes, h.vk_comm_unique_id.restype = [P], I
        h.vk_comm_init.argtypes, h.vk_comm_init.restype = [C.POINTER(P), P, I, I], I
        h.vk_comm_rank.argtypes, h.vk_comm_rank.restype = [P, C.POINTER(I), C.POINTER(I)], I
        h.vk_comm_allreduce_system.argtypes, h.vk_comm_allreduce_system.restype = [P, P, I, P], I
        h.vk_comm_reduce_hook.argtypes, h.vk_comm_reduce_hook.restype = [P, I, P, P], I
        h.vk_comm_count.argtypes, h.vk_comm_count.restype = [P, C.POINTER(I)], I
        h.vk_comm_exchange_attach.argtypes, h.vk_comm_exchange_attach.restype = [P, P], I
        h.vk_comm_exchange_detach.argtypes, h.vk_comm_exchange_detach.restype = [P, P], I
        h.vk_comm_exchange_create.argtypes, h.vk_comm_exchange_create.restype = [P, I, I, P], I
        h.vk_comm_exchange_attach_handles.argtypes, h.vk_comm_exchange_attach_handles.restype = [P, P], I
        h.vk_comm_exchange_next_sequence.argtypes, h.vk_comm_exchange_next_sequence.restype = [C.c_uint], C.c_uint
        h.vk_comm_destroy.argtypes, h.vk_comm_destroy.restype = [P], I
        h.vk_comm_error_string.argtypes, h.vk_comm_error_string.restype = [I], C.c_char_p
        _LIB = h
    return _LIB


def check(code, what):
    if code != 0:
        raise CommError(f"{what}: {lib().vk_comm_error_string(code).decode()} [{code}]")


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    check(lib().vk_comm_unique_id(buf), "vk_comm_unique_id")
    return buf.raw


class Communicator:
    def __init__(self, id_bytes, rank, world):
        self.handle = C.c_void_p()
        idbuf = None if id_bytes is None else C.create_string_buffer(bytes(id_bytes), ID_BYTES)
        check(lib().vk_comm_init(C.byref(self.handle), idbuf, rank, world), "vk_comm_init")
        self.rank, self.world = rank, world
        # the C function itself, passed as vk_icp_reduce_fn: nothing of Python runs per iteration
        self.hook_fn = C.cast(lib().vk_comm_reduce_hook, C.c_void_p)

    @classmethod
    def without_rccl(cls, rank, world):
        """A rank of a rig that exchanges only through attach_exchange_with (its own channel for the handles, the
        peer-mapped areas for the sums): no RCCL communicator, hence no all-reduce hook."""
        self = cls.__new__(cls)
        self.handle, self.rank, self.world, self.hook_fn = C.c_void_p(), rank, world, None
        return self

    @classmethod
    def from_torch_group(cls, rank, world):
        """Rank 0 creates the id; it travels through the default torch.distributed group."""
        if world == 1:
            return cls(None, 0, 1)
        import torch
        import torch.distributed as dist
        lib()
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
        raw = unique_id() if rank == 0 else bytes(ID_BYTES)
        t = torch.tensor(list(raw), dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0)
        return cls(bytes(t.cpu().tolist()), rank, world)

    def rccl_count(self):
        """Ranks RCCL itself reports for this communicator (ncclCommCount)."""
        n = C.c_int(-1)
        check(lib().vk_comm_count(self.handle, C.byref(n)), "vk_comm_count")
        return n.value

    def allreduce_system(self, system, stream=None):
        """In-place sum of a device float tensor over ranks on torch's current stream."""
        if stream is None and system.is_cuda:
            import torch
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib().vk_comm_allreduce_system(self.handle, C.c_void_p(system.data_ptr()), system.numel(), stream),
              "vk_comm_allreduce_system")
        return system

    def track(self, tracker, frame):
        tracker.comm = self
        return tracker.track(frame)

    # -- the exchange inside the one-launch loop (vk_rig_exchange, vk_icp_track_rig)
    exchange = None
    # Optional: `agree(flag) -> bool`, a collective "did ANY rank say True?" over the application's own channel (e.g.
    # agree_over_torch_group). With it, a Track that aborted on one rank raises TrackAborted on EVERY rank, so that
    # they can fall back together; without it only the rank that aborted raises, and the others' next collective
    # step would wait for it.
    agree = None

    def attach_exchange(self):
        """Collective: every rank's area mapped into every rank (vk_comm_exchange_attach); the outcome is the
        same on all ranks."""
        from . import vk_types as T
        x = T.RigExchange()
        check(lib().vk_comm_exchange_attach(self.handle, C.byref(x)), "vk_comm_exchange_attach")
        self.exchange = x
        return x

    def attach_exchange_with(self, all_gather, agree):
        """The same for ranks that move the IPC handles themselves (and may share a device, which RCCL refuses):
        `all_gather(bytes) -> [bytes per rank, in rank order]` and `agree(flag) -> bool` (True if any rank passed
        True) are the application's own collectives, e.g. torch.distributed over gloo."""
        from . import vk_types as T
        x = T.RigExchange()
        handle = C.create_string_buffer(IPC_HANDLE_BYTES)
        rc = lib().vk_comm_exchange_create(C.byref(x), self.rank, self.world, handle)
        if agree(rc != 0):                                   # a rank without an area: nobody goes on
            if rc == 0:
                lib().vk_comm_exchange_detach(None, C.byref(x))
            raise CommError(f"vk_comm_exchange_create failed on a rank (here: {rc})")
        handles = all_gather(handle.raw)
        assert len(handles) == self.world and all(len(b) == IPC_HANDLE_BYTES for b in handles)
        rc = lib().vk_comm_exchange_attach_handles(C.byref(x), C.create_string_buffer(b"".join(handles), IPC_HANDLE_BYTES * self.world))
        if agree(rc != 0):
            if rc == 0:
                lib().vk_comm_exchange_detach(None, C.byref(x))
            raise CommError(f"vk_comm_exchange_attach_handles failed on a rank (here: {rc})")
        self.exchange, self.agree = x, agree
        return x

    def track_rig(self, tracker, frame):
        """DepthTracker::Track on the rig with the ranks' sums exchanged inside the launch; every rank calls it
        for the same Track. The sequence number moves on HERE, on every rank, whether the Track ended with a pose or
        with VK_TRACK_ABORTED (vk_rig_protocol.h rig_next_sequence): a retry never meets the aborted attempt's words
        under their own tags, and the ranks' numbers cannot drift apart."""
        from . import api
        if self.exchange is None:
            self.attach_exchange()
        aborted, out, failure = False, None, None
        try:
            out = api.track_rig(tracker, frame, self.exchange)
        except api.TrackAborted:
            aborted = True
        except BaseException as e:     # noqa: BLE001  a VkError, a HIP error, a KeyboardInterrupt: anything that is not a pose
            # this rank takes no further part in the Track, and its peers must learn that the same way they learn of an
            # abort — otherwise they wait in agree(), or in their next in-launch exchange, for a rank that has left
            # (ADVICE r4). The exception travels on, after the collective.
            aborted, failure = True, e
        finally:
            self.exchange.sequence = lib().vk_comm_exchange_next_sequence(self.exchange.sequence)
        if self.agree is not None:
            try:
                aborted = bool(self.agree(aborted))          # collective: everybody learns of anybody's abort
            except BaseException:     # noqa: BLE001
                if failure is None:
                    raise
        if failure is not None:
            raise failure
        if aborted:
            raise api.TrackAborted("a rank of the rig gave up waiting for a peer's sums (VK_TRACK_ABORTED)")
        return out

    def detach_exchange(self):
        if self.exchange is not None:
            check(lib().vk_comm_exchange_detach(self.handle, C.byref(self.exchange)), "vk_comm_exchange_detach")
            self.exchange = None

    def time_allreduce(self, reps=200):
        """Microseconds per 48-float all-reduce, enqueued back to back."""
        import time
        import torch
        buf = torch.ones(48, dtype=torch.float32, device="cuda")
        for _ in range(10):
            self.allreduce_system(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.allreduce_system(buf)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    def close(self):
        self.detach_exchange()
        if self.handle:
            check(lib().vk_comm_destroy(self.handle), "vk_comm_destroy")
            self.handle = C.c_void_p()
