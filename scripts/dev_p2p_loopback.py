#!/usr/bin/env python3
"""Developer check for the one-GPU box: the RCCL point-to-point calls TileGather makes (batch_isend_irecv on a side
stream behind an event, variable sizes) with a single-rank group sending to itself."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    side = torch.cuda.Stream()
    for n in (1, 1000, 49 * 13300, 49 * 100000):
        x = torch.arange(n, dtype=torch.int32, device="cuda") * 3 + 1
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            y = torch.empty(n, dtype=torch.int32, device="cuda")
            ops = [dist.P2POp(dist.irecv, y, 0), dist.P2POp(dist.isend, x, 0)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            z = y + 0
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(z, x), n
        print("loopback ok", n, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
