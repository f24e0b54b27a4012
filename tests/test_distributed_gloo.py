"""World-size-2 test of the N>1 path on CPU (gloo): each process owns its Morton share of the
meshblocks, packs its halo messages according to the native plan, exchanges them with the
SAME HaloExchanger / all-reduce code the GPU path uses over RCCL, unpacks, and must end up
with exactly the oracle's whole-mesh ghost fill.  (Pack/unpack are done with numpy here --
on the GPU they are the copy_regions kernel.)"""
import os
import sys

import numpy as np
import pytest

from _spawn import spawn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _apply(regions, base):
    for reg in regions:
        src, dst = base(reg.src_kind, reg.src_block), base(reg.dst_kind, reg.dst_block)
        ii, jj, kk, vv = np.meshgrid(np.arange(reg.ext[0]), np.arange(reg.ext[1]), np.arange(reg.ext[2]),
                                     np.arange(reg.nvar), indexing="ij")
        so = reg.src_off + ii * reg.src_stride[0] + jj * reg.src_stride[1] + kk * reg.src_stride[2] + vv * reg.src_stride[3]
        do = reg.dst_off + ii * reg.dst_stride[0] + jj * reg.dst_stride[1] + kk * reg.dst_stride[2] + vv * reg.dst_stride[3]
        dst[do] = src[so]


def _worker(rank, world, port, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from athenapk_amd import decks, driver
    from oracle import oracle as O

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        if case == "mhd3d":
            deck, ov = "synthetic_mhd", ["parthenon/mesh/nx1=24", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16",
                                         "parthenon/meshblock/nx1=12", "parthenon/meshblock/nx2=8",
                                         "parthenon/meshblock/nx3=8"]
            o = O.Sim(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(24, 16, 16), mb=(12, 8, 8), ng=3)
            o.pgen("synthetic")
        else:
            deck, ov = "sod", ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8",
                               "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=4"]
            o = O.Sim(fluid="euler", recon="plm", riemann="hllc", integrator="rk2", nx=(32, 8, 8), mb=(8, 8, 4), ng=2,
                      bc=("outflow", "periodic", "periodic"), xmin=(0.0, -0.5, -0.5), xmax=(1.0, 0.5, 0.5), gamma=1.4)
            o.pgen("sod")
        rng = np.random.default_rng(42)  # same seed on both ranks: identical global start state
        for b in range(o.nblocks):
            o.cons(b)[...] = rng.uniform(1.0, 2.0, o.cons(b).shape)
        start = {b: o.cons(b).copy() for b in range(o.nblocks)}
        o.lib.orc_sim_exchange_ghosts(o.h)

        p = driver.HostPlan(decks.load(deck), ov, rank=rank, nranks=world)
        assert p.info.nblocks_local * world == o.nblocks
        blocks = [start[p.block_gid(lb)[0]].copy() for lb in range(p.info.nblocks_local)]
        peers = p.peers()
        assert [q for q, _, _ in peers] == [1 - rank]
        send = [torch.zeros(sc, dtype=torch.float64) for _, sc, _ in peers]
        recv = [torch.zeros(rc, dtype=torch.float64) for _, _, rc in peers]

        def base(kind, idx):
            if kind == 0:
                return blocks[idx].reshape(-1)
            return (send if kind == 1 else recv)[idx].numpy()

        _apply(p.regions("pack"), base)
        _apply(p.regions("local"), base)
        halo = driver.HaloExchanger([(q, send[n], recv[n]) for n, (q, _, _) in enumerate(peers)])
        halo.exchange()
        _apply(p.regions("unpack"), base)
        for ph in ("bc1", "bc2", "bc3"):
            _apply(p.regions(ph), base)
        for lb in range(p.info.nblocks_local):
            gid = p.block_gid(lb)[0]
            assert np.array_equal(blocks[lb], o.cons(gid)), "rank %d block %d" % (rank, gid)

        # the dt / c_h reductions (hydro.cpp:122-128): in-place MIN and SUM over ranks
        import ctypes as C
        vals = (C.c_double * 3)(1.0 + rank, 5.0 - rank, 7.0)
        driver._allreduce(vals, 3, dist.ReduceOp.MIN, torch.device("cpu"))
        assert list(vals) == [1.0, 4.0, 7.0]
        vals = (C.c_double * 2)(1.0 + rank, 0.5)
        driver._allreduce(vals, 2, dist.ReduceOp.SUM, torch.device("cpu"))
        assert list(vals) == [3.0, 1.0]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["mhd3d", "sod"])
def test_two_rank_halo_exchange_over_gloo(case):
    import torch.multiprocessing as mp
    spawn(_worker, lambda port: (2, port, case), 2)
