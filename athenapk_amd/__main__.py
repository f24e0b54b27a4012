"""`python -m athenapk_amd -i deck.in [-d outdir] [block/key=value ...]` -- runs an input deck
the way `athenaPK -i deck.in block/key=value` does for the scope of this package (uniform grid,
hydro / GLM-MHD flux-divergence update, history and linear-wave error outputs).  Under
`python -m torch.distributed.run --nproc-per-node N` it runs one rank per GPU over RCCL.
"""
import argparse
import ctypes as C
import os
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m athenapk_amd")
    ap.add_argument("-i", dest="deck", required=True, help="input deck (path, or the name of a deck in inputs/)")
    ap.add_argument("-d", dest="outdir", default=".", help="output directory")
    ap.add_argument("--strict", action="store_true", help="use the -ffp-contract=off build")
    ap.add_argument("overrides", nargs="*", help="block/key=value")
    a = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from . import decks, driver, lib as L

    if not torch.cuda.is_available():
        raise SystemExit("athenapk_amd needs a gfx950 GPU: there is no CPU fallback")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("APK_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("APK_DIST_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    text = open(a.deck).read() if os.path.exists(a.deck) else decks.load(a.deck)
    os.makedirs(a.outdir, exist_ok=True)
    sim = driver.Simulation(text, a.overrides, rank=rank, nranks=world, strict=a.strict)
    n = sim.execute(a.outdir)
    wall = sim.loop_seconds  # main loop only, like Parthenon's performance line
    if rank == 0:
        print("cycle=%d time=%.14e dt=%.14e" % (n, sim.time, sim.dt))
        print("zone-cycles/wallsecond = %.3e" % (sim.loop_zone_cycles / max(wall, 1e-30)))
    sim.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
