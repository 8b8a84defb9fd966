"""Diagnostic, run by hand (not collected by pytest):  python tests/gloo_device_probe.py
Two ranks on the one GPU of the box run the key replication with gloo collectives DIRECTLY on device
tensors (the key images wrapped from library-owned memory on rank 0): the form that once stalled.
Every rank arms faulthandler, so a stall prints the frame each rank sits in after 60 s and the
process exits; the parent waits with a bound.  The product never takes this route (its device-tensor
collectives are RCCL's); the two-rank test stages through the host."""
import faulthandler
import os
import socket
import sys
import time


def worker(rank, world, port, q):
    faulthandler.dump_traceback_later(60, exit=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from fhe_amd.sharding import max_over_ranks, replicated_context
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t0 = time.time()
    ctx, keep = replicated_context(dist, rank, world, dev, bytes(range(32)), 3)
    t1 = time.time()
    v = ctx.decrypt(ctx.evaluate("add", 8, ctx.encrypt(40 + rank, 8), ctx.encrypt(2, 8)))
    m = max_over_ranks(dist, 1.0 + rank, dev)
    q.put((rank, v, m, round(t1 - t0, 2)))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
    out = []
    while not q.empty():
        out.append(q.get())
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    print("ranks:", sorted(out), "exit codes:", codes, flush=True)
    sys.exit(0 if codes == [0, 0] and sorted(out)[0][1] == 42 and sorted(out)[1][1] == 43 else 1)
