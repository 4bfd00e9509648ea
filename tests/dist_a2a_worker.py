"""One rank over RCCL (the only RCCL a 1-GPU box can run): parallel._Coll.all_to_all_start / all_to_all_finish -- the overlapped form
the chunked payload exchange of merge2.merge_sharded_v2 uses -- against the synchronous all_to_all, with kernels issued in between
on the send buffer's neighbours, as the chunk loop does.  Launched by tests/test_merge2_gpu.py."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from avlmaps_amd import parallel  # noqa: E402


def main():
    rank, ws, local = parallel.init_distributed()
    assert ws == 1 and torch.distributed.get_backend() == "nccl"
    coll = parallel._Coll(None)
    g = torch.Generator(device="cuda").manual_seed(5)
    bufs = [torch.randint(0, 1 << 40, (1 << 18,), dtype=torch.int64, device="cuda", generator=g) for _ in range(2)]
    want = [b.clone() for b in bufs]
    got = []
    h = coll.all_to_all_start(bufs[0][:200_000], [200_000], [200_000])
    assert h[0] == "flying"
    for c in range(6):                                          # the chunk loop's order: refill the other buffer, finish, start the next, consume
        nxt = bufs[(c + 1) % 2]
        nxt.add_(c + 1)                                         # ("pack" of chunk c + 1: its previous exchange was finished an iteration ago)
        want[(c + 1) % 2] = want[(c + 1) % 2] + (c + 1)
        recv = coll.all_to_all_finish(h)
        h = coll.all_to_all_start(nxt[:200_000], [200_000], [200_000])
        got.append((recv.sum(), (want[c % 2][:200_000]).sum()))
    coll.all_to_all_finish(h)
    torch.cuda.synchronize()
    for a, b in got:
        assert int(a) == int(b)
    # empty exchanges and a switched-off overlap complete inside start
    assert coll.all_to_all_start(bufs[0][:0], [0], [0])[0] == "done"
    assert coll.all_to_all_start(bufs[0][:8], [8], [8], overlap=False)[0] == "done"
    sync = coll.all_to_all(bufs[0][:1000], [1000], [1000])
    assert torch.equal(sync, bufs[0][:1000]) and coll.calls >= 9
    print("A2A_OK", flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
