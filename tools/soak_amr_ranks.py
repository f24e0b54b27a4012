"""Development aid: the refined MHD blast of BASELINE config 5's shape on several ranks that share cuda:0 (gloo, messages
staged through the host), PRODUCT build, a few hundred cycles with regridding: every rank prints the global mass drift
(history() reduces over the ranks).   python tools/soak_amr_ranks.py [ranks=2] [cycles=400]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def worker(rank, world, port, ncyc):
    import numpy as np, torch
    import torch.distributed as dist
    from athenapk_amd import decks, driver
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ov = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + [
            "parthenon/mesh/numlevel=4", "parthenon/time/tlim=10.0", "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm",
            "parthenon/mesh/nghost=4", "problem/blast/pressure_ambient=1.0", "problem/blast/pressure_ratio=100"]
        s = driver.Simulation(decks.load("blast_3d_amr"), ov, rank=rank, nranks=world, strict=False).initialize()
        m0 = s.history()[0]
        for n in range(ncyc):
            s.step()
            if n % 100 == 99:
                h = s.history()[0]
                if rank == 0:
                    print("cycle", n + 1, "blocks", s.refresh_info().nblocks_total, "mass drift %.3e" % (abs(h - m0) / m0),
                          "passes skipped", s.amr_c2p_passes_skipped(), flush=True)
        s.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    from _spawn import spawn
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    spawn(worker, lambda port: (world, port, ncyc), world)
