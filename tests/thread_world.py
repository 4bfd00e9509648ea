"""parallel._Coll's interface over the threads of ONE process: the merge's choreography runs for several ranks without process groups
(CPU tensors with the NumPy twin of the kernels, or device tensors with the HIP kernels on one GPU)."""
import threading


class _NoLock:
    wait_s, held = 0.0, False


class ThreadWorld:
    def __init__(self, ws, exclusive=False):
        self.ws, self.barrier, self.slots = ws, threading.Barrier(ws), [None] * ws
        # exclusive: only ONE rank computes at a time (a lock taken at start, dropped inside every collective), and the time spent
        # inside collectives incl. waiting for the lock is booked to comm_s -- a rank's wall time minus comm_s is then what it computes
        # on a GPU of its own (tools/probe_merge2.py)
        self.lock = threading.Lock() if exclusive else None


class ThreadColl:
    """parallel._Coll's interface over threads of one process (device or CPU tensors)"""

    class _Dist:
        class ReduceOp:
            MAX = "max"

        @staticmethod
        def get_backend(group):
            return "threads"

    def __init__(self, world, rank):
        self.w, self.rank, self.ws = world, rank, world.ws
        self.comm_s, self.bytes_out, self.calls, self.gpu_lock, self.dist, self.group = 0.0, 0, 0, _NoLock(), self._Dist, None

    def _sync(self, t):
        if getattr(t, "is_cuda", False):
            import torch
            torch.cuda.synchronize()

    def _round(self, mine, take):
        import time
        self._sync(mine[0] if isinstance(mine, tuple) else mine)
        t0 = time.perf_counter()
        if self.w.lock is not None:
            self.w.lock.release()
        self.w.slots[self.rank] = mine
        self.w.barrier.wait()
        res = take(self.w.slots)
        self._sync(res[0] if isinstance(res, list) else res)
        self.w.barrier.wait()
        if self.w.lock is not None:
            self.w.lock.acquire()
        self.comm_s += time.perf_counter() - t0
        self.calls += 1
        return res

    def all_gather(self, t):
        return self._round(t, lambda s: [x.clone() for x in s])

    def all_gather_into(self, out, chunk):
        n = chunk.numel()

        def take(s):
            for r, c in enumerate(s):
                if r != self.rank:
                    out[r * n:(r + 1) * n].copy_(c)
            return out
        return self._round(chunk, take)

    def all_to_all(self, inp, in_splits, out_splits):
        import torch

        def take(s):
            parts = []
            for p, (t, ins) in enumerate(s):
                o = sum(ins[:self.rank])
                assert ins[self.rank] == out_splits[p], (p, self.rank, ins, out_splits)
                parts.append(t[o:o + ins[self.rank]])
            return torch.cat(parts) if parts else inp[:0]
        self.bytes_out += 8 * (sum(in_splits) - in_splits[self.rank])
        return self._round((inp, list(in_splits)), take)

    def all_to_all_start(self, inp, in_splits, out_splits, overlap=True):
        return ("done", self.all_to_all(inp, in_splits, out_splits))

    def all_to_all_finish(self, handle):
        return handle[1]

    def all_reduce(self, t, op):
        import torch
        return self._round(t, lambda s: torch.stack([x for x in s]).max(0).values)


def run_ranks(ws, fn, exclusive=False):
    world = ThreadWorld(ws, exclusive)
    out, errs = [None] * ws, []

    def body(r):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            if world.lock is not None:
                world.lock.acquire()
            try:
                out[r] = fn(r, ThreadColl(world, r))
            finally:
                if world.lock is not None and world.lock.locked():
                    try:
                        world.lock.release()
                    except RuntimeError:
                        pass
        except BaseException as e:      # noqa: BLE001 -- a failing rank must not leave the others in a barrier
            errs.append(e)
            world.barrier.abort()
    th = [threading.Thread(target=body, args=(r,)) for r in range(ws)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return out
