"""The native RCCL transport of the C++ host (csrc/host/comm_rccl.cpp): grouped ncclSend / ncclRecv per
peer on a dedicated HIP stream, ncclAllReduce on a second communicator.

* one GPU is enough for the self test: rank 0 sends to itself through the same code path (library
  lookup, two communicators, event ordering against the sim's stream, reductions);
* the 2-rank cases need two GPUs (RCCL refuses two ranks on one device) and are skipped on one-GPU
  boxes: they run the driver on device buffers over backend nccl with the native transport and must
  equal the oracle's single-process run bit for bit -- the same cases test_gpu_two_ranks.py runs over
  gloo with ranks sharing a GPU."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from _spawn import spawn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [1, 4097, 1 << 20])
def test_rccl_transport_self_test_on_one_gpu(n):
    import torch
    from athenapk_amd import lib as L
    assert torch.cuda.is_available()
    lib = L.load(strict=False)
    msg = C.create_string_buffer(256)
    rc = lib.apk_rccl_selftest(n, msg, len(msg))
    assert rc == L.APK_OK and msg.value == b"ok", msg.value


def _worker(rank, world, port, case, outdir, overlap):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from athenapk_amd import decks, driver
    from test_gpu_two_ranks import CASES
    torch.cuda.set_device(rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", rank))
    try:
        deck, ov, _, _, _, ncyc = CASES[case]
        s = driver.Simulation(decks.load(deck), ov, rank=rank, nranks=world, strict=True, comm="rccl")
        assert s.comm_kind == "rccl"
        s.set_overlap(overlap)
        s.initialize()
        for _ in range(ncyc):
            s.step()
        ex, red = C.c_longlong(0), C.c_longlong(0)
        s.lib.apk_sim_comm_stats(s.h, C.byref(ex), C.byref(red))
        blocks = {s.block_gid(lb)[0]: s.read_block(lb, "cons") for lb in range(s.info.nblocks_local)}
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), time=s.time, dt=s.dt, hist=s.history(), exchanges=ex.value,
                 reductions=red.value, overlapped=s.overlapped_exchanges, **{"b%d" % g: a for g, a in blocks.items()})
        s.close()
    finally:
        dist.destroy_process_group()


def _ngpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_ngpus() < 2, reason="RCCL needs one GPU per rank")
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("case", ["mhd_ppm_hlld_vl2", "mhd_ppm_two_kernel", "mhd_wenoz_hlld_rk3", "sod_outflow"])
def test_native_rccl_transport_two_gpus_matches_oracle(oracle, tmp_path, case, overlap):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_two_ranks import CASES
    deck, ov, okw, pgen, pkw, ncyc = CASES[case]
    o = oracle.Sim(nthreads=os.cpu_count(), **okw)
    o.pgen(pgen, **pkw)
    for _ in range(ncyc):
        o.step()
    spawn(_worker, lambda port: (2, port, case, str(tmp_path), overlap), 2)
    seen = set()
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert z["time"] == o.time and z["dt"] == o.dt
        assert int(z["exchanges"]) > 0 and int(z["reductions"]) > 0
        np.testing.assert_allclose(z["hist"], o.history(), rtol=1e-13, atol=1e-15)
        for key in z.files:
            if key.startswith("b"):
                seen.add(int(key[1:]))
                assert np.array_equal(z[key], o.cons(int(key[1:]))), "rank %d block %s" % (r, key)
    assert seen == set(range(o.nblocks))
