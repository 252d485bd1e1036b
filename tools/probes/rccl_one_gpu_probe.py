"""Can RCCL itself run between two ranks that share ONE GPU?  It refuses two ranks of one HOST on one device ("Duplicate GPU detected"), so each rank
declares a host of its own (NCCL_HOSTID) and the two meet over RCCL's socket transport on the loopback interface: RCCL's own point-to-point and
collective code, proxy threads and stream semantics — what the gloo runs of tests/test_gpu_multirank.py leave untested.
    timeout 180 python tools/probes/rccl_one_gpu_probe.py
"""
import datetime
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NCCL_HOSTID=f"flowmap-amd-probe-host-{rank}", NCCL_SOCKET_IFNAME="lo",
                      NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=60))
    out = {}
    x = torch.full((4,), float(rank + 1), device=dev)
    for name, fn in (("all_reduce", lambda: dist.all_reduce(x)),
                     ("all_gather", lambda: dist.all_gather([torch.zeros(2, device=dev) for _ in range(world)], torch.full((2,), float(rank), device=dev))),
                     ("broadcast", lambda: dist.broadcast(torch.full((3,), float(rank), device=dev), 0)),
                     ("batch_isend_irecv", lambda: [r.wait() for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, torch.ones(1 << 20, device=dev), 1 - rank),
                                                                                                dist.P2POp(dist.irecv, torch.zeros(1 << 20, device=dev), 1 - rank)])])):
        try:
            fn()
            torch.cuda.synchronize()
            out[name] = "ok"
        except Exception as exc:  # noqa: BLE001
            out[name] = f"{type(exc).__name__}: {str(exc)[:200]}"
    if rank == 0:
        print(out, "all_reduce ->", x.tolist(), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(run, args=(2, 29544), nprocs=2)
