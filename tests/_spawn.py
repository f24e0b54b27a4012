"""mp.spawn for the multi-process tests: a free port is found by binding port 0 and closing the socket, so another
process can take it before rank 0 of the spawned group listens on it (seen once in 800 GPU tests: EADDRINUSE).  The
group is then started again on another port -- nothing has run yet when the rendezvous fails."""
import socket


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn(fn, make_args, nprocs, attempts=3):
    """mp.spawn(fn, args=make_args(port), nprocs=nprocs, join=True), again with a fresh port if the port was taken"""
    import torch.multiprocessing as mp
    for attempt in range(attempts):
        try:
            mp.spawn(fn, args=make_args(free_port()), nprocs=nprocs, join=True)
            return
        except Exception as e:  # noqa: BLE001 -- ProcessRaisedException carries the child's traceback as text
            msg = str(e)
            if attempt + 1 < attempts and ("EADDRINUSE" in msg or "address already in use" in msg.lower()):
                continue
            raise
