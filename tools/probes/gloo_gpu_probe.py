"""Does this torch build's gloo backend take GPU tensors (two processes sharing ONE GPU)?  all_reduce / all_gather / broadcast / send-recv.
    python tools/probes/gloo_gpu_probe.py
"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def run(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    out = {}
    for name, fn in (("all_reduce", lambda: dist.all_reduce(torch.full((4,), float(rank + 1), device=dev))),
                     ("all_gather", lambda: dist.all_gather([torch.zeros(2, device=dev) for _ in range(world)], torch.full((2,), float(rank), device=dev))),
                     ("broadcast", lambda: dist.broadcast(torch.full((3,), float(rank), device=dev), 0)),
                     ("send_recv", lambda: (dist.send(torch.ones(2, device=dev), 1) if rank == 0 else dist.recv(torch.zeros(2, device=dev), 0))),
                     ("batch_isend_irecv", lambda: [r.wait() for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, torch.ones(2, device=dev), 1 - rank),
                                                                                                dist.P2POp(dist.irecv, torch.zeros(2, device=dev), 1 - rank)])])):
        try:
            fn()
            torch.cuda.synchronize()
            out[name] = "ok"
        except Exception as exc:  # noqa: BLE001
            out[name] = f"{type(exc).__name__}: {str(exc)[:120]}"
    if rank == 0:
        print(out, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(run, args=(2, 29533), nprocs=2)
