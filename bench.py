#!/usr/bin/env python
"""bench.py -- cell-updates/s (zone-cycles/s) of the 3-D GLM-MHD PPM+HLLD update on a uniform
grid (BASELINE.json metric), one process per GPU.

A "step" is one full cycle of the native driver on synthetic smooth MHD data resident in HBM:
c_h update -> for each VL2 stage [fused reconstruct->Riemann->flux-difference sweeps + RK update
+ Dedner source -> ghost-zone exchange -> ConsToPrim] -> hyperbolic dt estimate (+ min
all-reduce).  Weak scaling: every GPU owns a 256^3 brick = 8 meshblocks of 128^3; ghost zones
between bricks travel as one RCCL send/recv per peer per stage (torch.distributed "nccl").

  python bench.py                          # N=1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) including
  value        : the median of three timed regions of --steps cycles each (no event records inside)
  roofline     : algorithmic bytes of one fused stage / live HIP-event time of its kernels (a fourth region of the same
                 cycles with the library's event timing on), priced against HBM; `bound` names what binds the kernels;
                 general_stage.scheme_floor: the issue floor of the scheme's own arithmetic on this device
  rehearsal_8gpu_rank : one rank of the 2 x 2 x 2 run on this GPU (loopback transport), against the N = 1 brick timed beside it
  cpu_baseline : the CPU oracle (-O3 -march=native, OpenMP) on a bounded sample of the same
                 workload on this box's host cores -- a reported baseline, not the target.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
# HBM_PEAK_GBS: the MI355X HBM3E spec the roofline is priced against (MI355X_MICROARCH.md).
# HBM_COPY_GBS: what a float4 copy kernel reaches (same guide, HBM section); torch elementwise kernels on the
# bench boxes: copy 4950, read 6060, fill 6570 GB/s.
HBM_COPY_GBS = 6290.0

# algorithmic bytes (fp64, compulsory traffic) -- SURVEY.md 8(d)
B_STAGE = {"glmmhd": (216.0, 288.0), "euler": (120.0, 160.0)}   # per cell-stage: gam0 == 0 / != 0
B_C2P = {"glmmhd": 144.0, "euler": 80.0}                        # ConsToPrim per cell-stage
GAM0 = {"rk1": (0.0,), "rk2": (0.0, 0.5), "vl2": (0.0, 0.0), "rk3": (0.0, 0.25, 2.0 / 3.0)}

WORKLOADS = {
    # name: (deck, fluid, integrator, recon, riemann, per-GPU brick, meshblock, description)
    # default = the configuration the BASELINE metric is quoted on (3-D MHD PPM+HLLD, uniform grid)
    "mhd_ppm_hlld_vl2_256": ("synthetic_mhd", "glmmhd", "vl2", "ppm", "hlld", 256, 128,
                             "GLM-MHD PPM+HLLD+Dedner VL2, synthetic smooth state, 256^3 per GPU in 128^3 meshblocks"),
    # BASELINE configs[3] scheme (driven-turbulence config without the forcing term): WENOZ+HLLD RK3
    "mhd_wenoz_hlld_rk3_256": ("synthetic_mhd", "glmmhd", "rk3", "wenoz", "hlld", 256, 128,
                               "GLM-MHD WENOZ+HLLD+Dedner RK3, synthetic smooth state, 256^3 per GPU in 128^3 meshblocks"),
    # BASELINE configs[1]: Sod shock tube 256^3, hydro PLM+HLLC RK2
    "hydro_plm_hllc_rk2_256": ("sod", "euler", "rk2", "plm", "hllc", 256, 128,
                               "hydro PLM+HLLC RK2, 3-D Sod shock tube (outflow x1), 256^3 per GPU in 128^3 meshblocks"),
}
RANK_GRID = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


def _usable_cores():
    """host cores this process may really use: affinity mask and cgroup CPU quota, not the node size"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def _oracle_rate(fluid, integrator, recon, riemann, n, mb, threads, budget_s):
    from oracle import oracle as O
    sim = O.Sim(fluid=fluid, recon=recon, riemann=riemann, integrator=integrator, nx=(n, n, n), mb=(mb, mb, mb),
                ng=3 if recon in ("ppm", "wenoz") else 2, xmax=(1.0, 1.0, 1.0), cfl=0.3, nthreads=threads, fast=True)
    sim.pgen("synthetic")
    t0 = time.perf_counter()
    sim.step()
    t1 = time.perf_counter() - t0
    cycles = int(min(50, max(1, math.floor(budget_s / max(t1, 1e-3)))))
    t0 = time.perf_counter()
    for _ in range(cycles):
        sim.step()
    dt = time.perf_counter() - t0
    return n ** 3 * cycles / dt, cycles, dt


def cpu_baseline(fluid, integrator, recon, riemann, target_s=10.0):
    """Times the oracle (kind 'port': no reference binary can be built here) on the host cores of
    this box: (i) one thread -- the analogue of the reference's Kokkos-serial build -- and (ii) the
    OpenMP build on the thread count that runs fastest (node-sized thread counts lose on boxes
    whose container is pinned to fewer cores)."""
    cores = _usable_cores()
    serial, _, _ = _oracle_rate(fluid, integrator, recon, riemann, 64, 32, 1, 2.0)
    cands = sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores} | {min(cores, 8)})
    best_t, best = cands[0], 0.0
    for t in cands:  # quick calibration, ~1 s each
        r, _, _ = _oracle_rate(fluid, integrator, recon, riemann, 64, 16, t, 0.5)
        if r > best:
            best_t, best = t, r
    n, mb = 128, (16 if best_t > 64 else 32)
    value, cycles, dt = _oracle_rate(fluid, integrator, recon, riemann, n, mb, best_t, target_s)
    return {"value": value, "unit": "cell-updates/s", "cores": best_t, "kind": "port",
            "serial_value": serial, "usable_cores": cores,
            "sample": "%d cycles of the same scheme on a %d^3 mesh in %d^3 meshblocks with %d OpenMP threads "
                      "(fastest of %s; %d usable cores), oracle built -O3 -march=native -fopenmp, %.1f s; "
                      "serial_value: 1 thread on 64^3" % (cycles, n, mb, best_t, cands, cores, dt)}


MIN_REGION_MS = 50.0  # every driver-visible rate: the median of three timed regions at least this long


def timed_regions(step, sync, probe_cycles=4, regions=3, min_ms=MIN_REGION_MS, min_cycles=4):
    """`regions` timed regions of the same number of cycles, each at least `min_ms` long (sized from a short probe);
    returns (cycles per region, [seconds per region], index of the median region).  The reference's performance suite
    reads the rate Parthenon prints over a whole run (tst/regression/test_suites/performance/performance.py:32-54);
    a region of a few milliseconds cannot separate a 2 % change from the box."""
    sync()
    t0 = time.perf_counter()
    for _ in range(probe_cycles):
        step()
    sync()
    per = (time.perf_counter() - t0) / probe_cycles
    cycles = max(min_cycles, int(math.ceil(1.1 * min_ms * 1e-3 / per)))
    out = []
    while len(out) < regions:
        sync()
        t0 = time.perf_counter()
        for _ in range(cycles):
            step()
        sync()
        e = time.perf_counter() - t0
        if e * 1e3 < min_ms:  # (the probe overestimated a cycle -- a rejected trial stage, a regridding: longer regions, start over)
            cycles = int(math.ceil(cycles * 1.25 * min_ms * 1e-3 / e))
            out = []
            continue
        out.append(e)
    med = sorted(range(len(out)), key=lambda q: out[q])[len(out) // 2]
    return cycles, out, med


def amr_blast_bench(cycles=40, variants=("hydro_plm_hlle_vl2", "mhd_ppm_hlld_vl2")):
    """BASELINE config 5's shape (inputs/blast_3d_amr.in with root 64^3 in 16^3 meshblocks, 4 levels,
    regridding every cycle) for the hydro deck as it is and for GLM-MHD PPM+HLLD on the deck's own blast
    (near-vacuum ambient medium, pressure ratio 1.6e8, no first-order flux correction: the deck has none and
    tests/test_gpu_configs.py::test_config5_adaptive_mhd_blast_as_decked runs it for 400 cycles without a
    negative state): zone-cycles/s counted over the blocks that exist in each cycle, like Parthenon's
    performance line.  Supplementary."""
    import torch
    from athenapk_amd import decks, driver
    ov = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)]
    ov += ["parthenon/mesh/numlevel=4"]
    out = {}
    for name, extra in (("hydro_plm_hlle_vl2", []),
                        ("mhd_ppm_hlld_vl2", ["hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm",
                                              "parthenon/mesh/nghost=4"])):
        if name not in variants:
            continue
        s = driver.Simulation(decks.load("blast_3d_amr"), ov + extra).initialize()
        for _ in range(3):
            s.step()
        # (the mesh grows with the blast: the regions are consecutive stretches of the same run, zone-cycles counted over
        # the blocks that exist in each cycle; the median RATE is reported)
        rates, times, per_cycle = [], [], []
        for _ in range(3):
            torch.cuda.synchronize()
            z0, t0 = s.amr_stats()[3], time.perf_counter()
            n = 0
            while True:
                for _ in range(8):
                    s.step()
                n += 8
                torch.cuda.synchronize()
                if n >= cycles and (time.perf_counter() - t0) * 1e3 >= 1.05 * MIN_REGION_MS:
                    break
            dt = time.perf_counter() - t0
            rates.append((s.amr_stats()[3] - z0) / dt)
            times.append(dt * 1e3)
            per_cycle.append(dt / n * 1e3)
        med = sorted(range(3), key=lambda q: rates[q])[1]
        refined, merged, maxlev, z1 = s.amr_stats()
        i = s.refresh_info()
        out[name] = {"zone_cycles_per_s": rates[med], "ms_per_cycle": per_cycle[med], "timed_regions_ms": times,
                     "zone_cycles_per_s_of_the_regions": rates,
                     "meshblocks": int(i.nblocks_total), "levels": maxlev + 1, "blocks_refined": refined,
                     "sibling_groups_merged": merged}
        s.close()
    out["mesh"] = "root 64^3 in 16^3 meshblocks, 4 levels, adaptive (pressure gradient), fused stages + post-stage flux correction"
    return out


def scheme_floor(cells, stage_ms, steps=256, reps=8):
    """The empirical issue floor of the PPM + HLLD scheme on this device (csrc/bench_floor.hip through the C-ABI): every lane of
    two waves per SIMD -- the stage kernels' occupancy -- runs `steps` sweep steps of nine PPM reconstructions + one HLLD
    solve on a smooth monotone pencil in registers / its LDS ring and NOTHING else (no global memory, no wave shifts, no
    limiter block entered, no update / Dedner / ConsToPrim); a cell-stage of the 3-D scheme is three such steps."""
    import ctypes as C
    from athenapk_amd import lib as L
    lib = L.load(False)
    ms, n = C.c_double(0.0), C.c_longlong(0)
    rc = lib.apk_bench_scheme_floor(steps, reps, C.byref(ms), C.byref(n))
    if rc != L.APK_OK:
        return {"error": "apk_bench_scheme_floor: %d" % rc}
    per_step_ns = ms.value * 1e6 / n.value          # ns per lane-step with the whole device busy
    floor_ms = 3.0 * cells * per_step_ns * 1e-6
    return {"what": "9 PPM reconstructions + 1 HLLD solve per sweep step, nothing else, two waves per SIMD (csrc/bench_floor.hip); "
                    "floor_ms_per_stage = 3 steps per cell x the cells of the stage benchmark",
            "lane_steps_per_s": n.value / (ms.value * 1e-3), "floor_ms_per_stage": floor_ms,
            "frac_of_scheme_floor": floor_ms / stage_ms,
            "north_star_40pct_ms_per_stage": 288.0 * cells / (0.40 * HBM_PEAK_GBS * 1e9) * 1e3}


def general_stage_bench(recon="ppm", riemann="hlld", nb=8, n=128, reps=5):
    """SURVEY 8(d) "synthetic kernel benchmark (north_star target)": one pack of nb random-smooth
    128^3 GLM-MHD blocks, the GENERAL RK stage (gam0 = gam1 = 1/2: u0 is read as well, 288 B per
    cell-stage), timed with events on the stream the kernels run on, through the C-ABI."""
    import torch
    from athenapk_amd import hydro, lib as L
    ng, dev = 3, torch.device("cuda")
    N = n + 2 * ng
    ctx = hydro.Context(strict=False)
    ax = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 * math.pi / n)
    k, j, i = torch.meshgrid(ax, ax, ax, indexing="ij")
    w = torch.empty((nb, 9, N, N, N), dtype=torch.float64, device=dev)
    for b in range(nb):
        ph = 0.37 * b
        w[b, 0] = 1.0 + 0.2 * torch.sin(i + 2 * j + k + ph)
        w[b, 1] = 0.3 * torch.sin(j - k + ph)
        w[b, 2] = 0.3 * torch.cos(i + k)
        w[b, 3] = 0.3 * torch.sin(i - 2 * j + ph)
        w[b, 4] = 1.0 + 0.1 * torch.cos(2 * i + j - k)
        w[b, 5] = 0.5 * torch.sin(j + ph)
        w[b, 6] = 0.5 * torch.cos(k - i)
        w[b, 7] = 0.5 * torch.sin(i + j + ph)
        w[b, 8] = 0.01 * torch.sin(i + j + k)
    gamma = 5.0 / 3.0
    u = w.clone()
    u[:, 1:4] = w[:, 0:1] * w[:, 1:4]
    u[:, 4] = (w[:, 4] / (gamma - 1.0) + 0.5 * w[:, 0] * (w[:, 1:4] ** 2).sum(1) + 0.5 * (w[:, 5:8] ** 2).sum(1)
               + 0.5 * w[:, 8] ** 2)
    dx = (1.0 / n,) * 3
    m0 = hydro.MeshData(ctx, (n, n, n), ng, 9, dx=dx, nblocks=nb, cons=u, prim=w, with_flux=False)
    m1 = hydro.MeshData(ctx, (n, n, n), ng, 9, dx=dx, nblocks=nb, cons=u.clone(), with_flux=False)  # u1 != u0
    del u, w
    eos = L.make_eos(gamma)

    def stage():
        hydro.StageFused(m0, m1, "glmmhd", recon, riemann, eos, 2.0, 0.5, 0.5, 1e-7, dedner=1, glmmhd_alpha=0.1,
                         mindx=dx[0])
    for _ in range(2):
        stage()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)   # hydro.StageFused launches on torch's current stream
    for _ in range(reps):
        stage()
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    cells = nb * n ** 3
    gbs = 288.0 * cells / (ms * 1e-3) / 1e9
    floor = None
    if recon == "ppm" and riemann == "hlld":
        floor = scheme_floor(cells, ms)
    return {"scheme_floor": floor,
            "description": "one pack of %d smooth %d^3 GLM-MHD blocks, %s+%s general stage (gam0 = gam1 = 1/2), "
                           "288 B per cell-stage (SURVEY 8(d))" % (nb, n, recon.upper(), riemann.upper()),
            "ms_per_stage": ms, "cell_stage_updates_per_s": cells / (ms * 1e-3), "achieved": gbs, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "frac_of_measured_copy_bandwidth": gbs / HBM_COPY_GBS}


# rocprofv3 names of the kernels behind the driver's timing slots (3-D two-kernel stage; template arguments are
# <fluid, recon, riemann, extra, lean> with the enum values of include/apk_amd.h)
_FLUID_ID = {"euler": 1, "glmmhd": 2}
_RECON_ID = {"dc": 1, "plm": 2, "ppm": 3, "wenoz": 4, "weno3": 5, "limo3": 6}
_RIEMANN_ID = {"hlle": 2, "llf": 3, "hllc": 4, "hlld": 5}


def rocprof_kernels(fluid, recon, riemann):
    f, r, s = _FLUID_ID[fluid], _RECON_ID[recon], _RIEMANN_ID[riemann]
    return {"fused_x1": "fused_m12f_kernel<%d, %d, %d, 2, 1, FC, X1H> / <%d, %d, %d, 0, 1, FC, X1H> (x1 + x2 finishing march; "
                        "last stage of a cycle with ConsToPrim + dt / other stages; 1 = the lean form; FC = true where the stage "
                        "derives its input from the conserved state: the RK integrators; X1H = true where x1 strips live in the "
                        "exchange buffers: N > 1)" % (f, r, s, f, r, s),
            "fused_x3": "fused_march_kernel<%d, %d, %d, 3, false, 0, FC> (x3 sweep)" % (f, r, s),
            "fused_dc_x1": "fused_dc3r2_kernel<%d, %d, 1, true, X1H, 1> (donor-cell predictor stage, two rows per lane, input derived from the "
                           "conserved state; <.., false, ..> in the first cycle)" % (f, s)}


ROCPROF_KERNEL = rocprof_kernels("glmmhd", "ppm", "hlld")


def stage_figures(timing, fluid, integrator, zones_local, ndim=3):
    """per-kernel averages and the high-order stage's roofline figures from the driver's HIP-event timing slots"""
    per_kernel = {k: (ms / n if n else 0.0) for k, (ms, n) in timing.items()}
    fin = "fused_x3" if ndim == 3 else "fused_x2"
    if timing[fin][1]:  # (with the exchange overlapped the first sweep of a stage is several launches)
        per_kernel["fused_x1"] = timing["fused_x1"][0] / timing[fin][1]
    stage_ms = per_kernel["fused_x1"] + per_kernel["fused_x2"] + per_kernel["fused_x3"]
    ho = [g0 for n, g0 in enumerate(GAM0[integrator]) if not (integrator == "vl2" and n == 0)]
    b_stage = sum(B_STAGE[fluid][0 if g0 == 0.0 else 1] for g0 in ho) / len(ho)
    achieved = b_stage * zones_local / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
    dominant = max(per_kernel, key=lambda k: timing[k][0])
    return per_kernel, stage_ms, b_stage, achieved, dominant


# Workloads that only appear under `other_workloads` (never the headline): BASELINE configs[3] WITH its forcing
# (inputs/turbulence.in: 30 modes, the deck's amplitudes; the scheme of the config) and configs[2], the Orszag-Tang
# vortex 512 x 512 thin in x3 with the deck's first_order_flux_correct off and on (inputs/orszag_tang.in:15-51).
# (deck, fluid, integrator, recon, riemann, mesh, meshblock, description, extra overrides)
EXTRA_WORKLOADS = {
    "mhd_wenoz_hlld_rk3_256_forced": ("turbulence", "glmmhd", "rk3", "wenoz", "hlld", (256, 256, 256), (128, 128, 128),
                                      "BASELINE config 4 as specified: GLM-MHD WENOZ+HLLD RK3 with the few-modes forcing on "
                                      "(inputs/turbulence.in), 256^3 per GPU in 128^3 meshblocks",
                                      ["parthenon/mesh/nghost=3"]),
    "orszag_tang_512x512x4_vl2": ("orszag_tang", "glmmhd", "vl2", "ppm", "hlld", (512, 512, 4), (128, 128, 4),
                                  "BASELINE config 3: Orszag-Tang 512 x 512 x 4 (thin-z 3-D), GLM-MHD PPM+HLLD+Dedner VL2, "
                                  "first_order_flux_correct off", ["hydro/first_order_flux_correct=false"]),
    "orszag_tang_512x512x4_vl2_fofc": ("orszag_tang", "glmmhd", "vl2", "ppm", "hlld", (512, 512, 4), (128, 128, 4),
                                       "BASELINE config 3 as decked: the same with first_order_flux_correct on",
                                       ["hydro/first_order_flux_correct=true"]),
    # ... and where the correction has work to do: the same run advanced to t = 0.94 in warm-up (3500 cycles, 1.4 s), where
    # the strongest shocks meet -- a fraction of the cycles has a trial stage rejected (the stage redone through the flux
    # arrays after its ghost zones were filled) and a few cells take first-order fluxes (hydro.cpp:1223-1342)
    "orszag_tang_512x512x4_vl2_fofc_late": ("orszag_tang", "glmmhd", "vl2", "ppm", "hlld", (512, 512, 4), (128, 128, 4),
                                            "BASELINE config 3 as decked, timed from t = 0.94 on (the vortex's shocks have formed): "
                                            "first-order flux correction with rejected trial stages and corrected cells",
                                            ["hydro/first_order_flux_correct=true"]),
}
WARM_UNTIL_TIME = {"orszag_tang_512x512x4_vl2_fofc_late": 0.94}


def other_workload(name, steps=4, warmup=2):
    """One of the non-headline workloads on this GPU, a few cycles: rate, ms per cycle, the high-order
    stage against the HBM roofline and the dominant kernel (the reference's own performance suite is a matrix of
    schemes, tst/regression/test_suites/performance/performance.py:32-54)."""
    import torch
    from athenapk_amd import decks, driver
    if name in EXTRA_WORKLOADS:
        deck, fluid, integrator, recon, riemann, mesh, mbs, desc, extra = EXTRA_WORKLOADS[name]
    else:
        deck, fluid, integrator, recon, riemann, brick, mb, desc = WORKLOADS[name]
        mesh, mbs, extra = (brick,) * 3, (mb,) * 3, []
    ov = ["parthenon/mesh/nx%d=%d" % (d + 1, mesh[d]) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mbs[d]) for d in range(3)]
    ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann] + extra
    sim = driver.Simulation(decks.load(deck), ov, strict=False).initialize()
    try:
        for _ in range(warmup):
            sim.step()
        while name in WARM_UNTIL_TIME and sim.time < WARM_UNTIL_TIME[name] and sim.ncycle < 20000:
            sim.step()
        fofc0 = (sim.ncycle, sim.fofc_count, sim.fofc_fallback_stages)
        steps, regions, med = timed_regions(sim.step, torch.cuda.synchronize, probe_cycles=max(2, steps // 2))
        dt = regions[med]
        fofc1 = (sim.ncycle, sim.fofc_count, sim.fofc_fallback_stages)
        # per-kernel averages from a few more cycles with the in-library event timing on: two event records per launch
        # are 10 % of the sub-millisecond cycles of the small workloads (config 3), so they stay out of the timed loop
        sim.kernel_timing(True)
        sim.read_kernel_timing()
        for _ in range(max(2, min(16, steps // 2))):
            sim.step()
        torch.cuda.synchronize()
        timing = sim.read_kernel_timing()
        sim.kernel_timing(False)
        info = sim.info
        per_kernel, stage_ms, b_stage, achieved, dominant = stage_figures(timing, fluid, integrator, int(info.zones_local), info.ndim)
        out = {"description": desc, "value": int(info.zones_total) * steps / dt, "unit": "cell-updates/s", "steps": steps,
               "ms_per_step": dt / steps * 1e3, "timed_regions_ms": [e * 1e3 for e in regions],
               "timed_region_reported": "median of %d regions of %d cycles each" % (len(regions), steps),
               "high_order_stage_ms": stage_ms,
               "dc_predictor_stage_ms": per_kernel["fused_dc_x1"] + per_kernel["fused_dc_x2"] + per_kernel["fused_dc_x3"],
               "algorithmic_bytes_per_cell_stage": b_stage, "stage_achieved_GBps": achieved, "stage_frac": achieved / HBM_PEAK_GBS,
               "dominant_timing_slot": dominant,
               "dominant_kernel": (rocprof_kernels(fluid, recon, riemann).get(dominant, dominant) if info.ndim == 3 and min(mbs) >= 16 else dominant),
               "per_kernel_avg_ms": {k: v for k, v in per_kernel.items() if v > 0.0}}
        if deck == "orszag_tang":
            # FirstOrderFluxCorrect (hydro.cpp:1223-1342): cells whose fluxes were replaced / stages whose optimistic fused
            # form was rejected and redone on the flux arrays, per cycle of the timed regions (early on the vortex is
            # smooth and the option costs its admissibility test only; the `_late` workload times it where it fires)
            ncyc = max(1, fofc1[0] - fofc0[0])
            out["fofc_cells_corrected_per_cycle"] = (fofc1[1] - fofc0[1]) / ncyc
            out["fofc_fallback_stages_per_cycle"] = (fofc1[2] - fofc0[2]) / ncyc
            out["simulation_time_at_the_start_of_the_timed_regions"] = None if name not in WARM_UNTIL_TIME else WARM_UNTIL_TIME[name]
        if deck == "turbulence":
            out["forcing_kicks_without_stored_primitives"] = sim.turb_dt_kicks()
        return out
    finally:
        sim.close()


XGMI_LINK_GBS = 153.0  # one xGMI link per peer GPU (MI355X_MICROARCH.md: 7 links x ~153 GB/s per direction)


def rehearsal_8gpu_rank(workload, n1_ms, steps=10, warmup=2):
    """One-GPU rehearsal of ONE RANK of the 8-GPU run of `workload` (the driver runs the real curve; a one-GPU box
    cannot): the same 256^3 brick with the three outer faces of every meshblock -- and the brick's edges and corner --
    treated as faces to other ranks (deck parameter apk_amd/rehearse_remote_faces): pack kernel -> one message per
    peer on the halo stream -> unpack + ghost ConsToPrim, overlapped with the next stage's plane windows exactly as
    exchange_begin / exchange_end do between bricks (hydro_driver.cpp:506, 567-568).  The seven messages are delivered
    by device copies on the halo stream (loopback transport, csrc/host/comm_rccl.cpp) instead of ncclSend / ncclRecv
    over xGMI: everything but the wire is timed; the wire time is modelled next to it."""
    import torch
    from athenapk_amd import decks, driver
    deck, fluid, integrator, recon, riemann, brick, mb, desc = WORKLOADS[workload]
    ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
    ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann]
    out = {}
    # the three variants back to back in this function, each the median of three regions of >= 50 ms without event records:
    # the plain N = 1 brick again (the reference the exposed time is counted from -- the headline's own regions ran
    # minutes earlier, on other clocks), then the rehearsal with its exchanges overlapped and synchronous
    for label, overlap, extra in (("n1", True, []), ("overlapped", True, ["apk_amd/rehearse_remote_faces=true"]),
                                  ("synchronous", False, ["apk_amd/rehearse_remote_faces=true"])):
        sim = driver.Simulation(decks.load(deck), ov + extra, strict=False)
        sim.set_overlap(overlap)
        sim.initialize()
        try:
            for _ in range(warmup):
                sim.step()
            ov0, thin0, x10 = sim.overlapped_exchanges, sim.thin_exchanges(), sim.x1_direct_exchanges()
            cyc, regions, med = timed_regions(sim.step, torch.cuda.synchronize, probe_cycles=3)
            dt = regions[med]
            done = 3 + 3 * cyc
            out[label] = {"ms_per_step": dt / cyc * 1e3, "timed_regions_ms": [e * 1e3 for e in regions], "cycles_per_region": cyc,
                          "value": int(sim.info.zones_total) * cyc / dt}
            if label == "n1":
                continue
            # (the copy kernels' HIP-event durations from a few more cycles with the in-library timing on)
            sim.kernel_timing(True)
            sim.read_kernel_timing()
            for _ in range(steps):
                sim.step()
            torch.cuda.synchronize()
            timing = sim.read_kernel_timing()
            sim.kernel_timing(False)
            peers = sim.messages("uniform")
            thin_peers = sim.messages("uniform_thin")
            out[label].update({"one_layer_exchanges_per_cycle": (sim.thin_exchanges() - thin0) / (done + steps),
                               "overlapped_exchanges_per_cycle": (sim.overlapped_exchanges - ov0) / (done + steps),
                               "exchanges_with_x1_strips_in_the_buffers_per_cycle": (sim.x1_direct_exchanges() - x10) / (done + steps),
                               "pack_unpack_copy_kernels_ms_per_cycle": timing["copy_regions"][0] / steps})
        finally:
            sim.close()
    headline_ms, n1_ms = n1_ms, out["n1"]["ms_per_step"]
    nst = len(GAM0[integrator])
    msg = sorted((8.0 * sc for _, sc, _ in peers), reverse=True)
    msg_thin = sorted((8.0 * sc for _, sc, _ in thin_peers), reverse=True)
    wire_ms = msg[0] / (XGMI_LINK_GBS * 1e9) * 1e3  # the three face messages travel on three links at once
    wire_thin_ms = msg_thin[0] / (XGMI_LINK_GBS * 1e9) * 1e3
    ms = out["overlapped"]["ms_per_step"]
    n_thin = out["overlapped"]["one_layer_exchanges_per_cycle"]
    n_over = out["overlapped"]["overlapped_exchanges_per_cycle"]
    # Exchanges left in flight behind the next stage's first kernel hide their wire time there (or not: second figure);
    # the others -- the one before the donor-cell predictor, which runs whole -- pay it in full.  The one-layer exchange is
    # the one before the predictor: never hidden.
    wire_all = (nst - n_thin) * wire_ms + n_thin * wire_thin_ms
    wire_exposed = max(0.0, nst - n_thin - n_over) * wire_ms + n_thin * wire_thin_ms
    out.update({
        "what": "one rank of the 2 x 2 x 2 run rehearsed on one GPU: 7 peers (3 faces, 3 edges, 1 corner), messages delivered by "
                "ONE copy kernel on the halo stream (loopback) instead of xGMI; n1_ms_per_step is the plain N = 1 brick timed in the same "
                "function right before (three regions of >= 50 ms each, median, like the two rehearsals)",
        "n1_ms_per_step": n1_ms,
        "headline_ms_per_step": headline_ms,
        "exchanges_per_cycle": nst,
        "message_MB_per_peer": [m / 1e6 for m in msg],
        "one_layer_message_MB_per_peer": [m / 1e6 for m in msg_thin],
        "exposed_exchange_ms_per_cycle": ms - n1_ms,
        "exposed_exchange_ms_per_cycle_without_overlap": out["synchronous"]["ms_per_step"] - n1_ms,
        "wire_ms_per_exchange_modelled": wire_ms,
        "wire_ms_per_one_layer_exchange_modelled": wire_thin_ms,
        "wire_model": "largest message / %.0f GB/s (one xGMI link per face peer, the three face messages on three links at once; "
                      "edge and corner messages are 1 - 3 %% of a face's)" % XGMI_LINK_GBS,
        # MODEL on a LOOPBACK rehearsal, not a measurement: N = 1 time / (rehearsed rank time + modelled exposed wire time)
        "predicted_weak_scaling_efficiency": n1_ms / (ms + wire_exposed),
        "predicted_weak_scaling_efficiency_if_no_wire_time_is_hidden": n1_ms / (ms + wire_all),
        "predicted_weak_scaling_efficiency_is": "a model on a one-GPU loopback rehearsal (no xGMI link was used): N = 1 ms / (rehearsal ms + "
                                                "modelled exposed wire ms); the measured curve is the driver's SCALE file",
    })
    return out


def other_workloads():
    """BASELINE configs 2, 3 (first-order flux correction off and on), 4 (its scheme unforced, and as specified with the
    forcing) and 5 (its mesh) next to the headline: a few cycles each, after the headline's timed regions so that nothing
    perturbs them."""
    out = {}
    for name in ("hydro_plm_hllc_rk2_256", "mhd_wenoz_hlld_rk3_256", "mhd_wenoz_hlld_rk3_256_forced",
                 "orszag_tang_512x512x4_vl2", "orszag_tang_512x512x4_vl2_fofc", "orszag_tang_512x512x4_vl2_fofc_late"):
        try:
            out[name] = other_workload(name)
        except Exception as e:  # supplementary figures; never lose the headline
            out[name] = {"error": repr(e)}
    try:
        out["amr_blast_cfg5_mesh"] = amr_blast_bench(cycles=40, variants=("mhd_ppm_hlld_vl2",))
    except Exception as e:
        out["amr_blast_cfg5_mesh"] = {"error": repr(e)}
    return out


FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz; MI355X_MICROARCH.md)
N_SIMD = 1024                  # 256 CUs x 4
FP64_ISSUE_CYCLES = 4.0        # a wave64 fp64 instruction occupies its SIMD's 16 lanes for 4 cycles


def valu_roofline(stage_ms):
    """The compute-side roof SURVEY 8(d) asks for next to the HBM one (the fp64 vector ALU), for the general PPM+HLLD
    stage: fp64 add / mul / fma wave-instructions per stage from the newest COMMITTED counter profile of the stage
    benchmark (profiles/rNN_pmc_instruction_mix.json: SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 of the two stage kernels, product
    build), priced at full width (64 lanes, an fma = 2 flop) against `stage_ms` measured live in this run."""
    for rnd in ("r06", "r05", "r04"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_instruction_mix.json" % rnd)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                mix = json.load(f)["mix"]
            add = sum(k["SQ_INSTS_VALU_ADD_F64"] for k in mix.values())
            mul = sum(k["SQ_INSTS_VALU_MUL_F64"] for k in mix.values())
            fma = sum(k["SQ_INSTS_VALU_FMA_F64"] for k in mix.values())
            valu = sum(k["SQ_INSTS_VALU"] for k in mix.values())
            # live lanes per VALU instruction: SQ_THREAD_CYCLES_VALU counts them per PASS through the pipe, and an fp64
            # transcendental makes four passes (a plain quotient can exceed 64: round-5 review)
            trans = sum(k.get("SQ_INSTS_VALU_TRANS_F64", 0.0) for k in mix.values())
            lanes = sum(k["SQ_THREAD_CYCLES_VALU"] for k in mix.values()) / (valu + 3.0 * trans)
            flops = 64.0 * (add + mul + 2.0 * fma)
            tf = flops / (stage_ms * 1e-3) / 1e12
            commit = _profile_commit(path)
            return {"kernel": "general PPM+HLLD stage (x3 sweep + finishing march), the `general_stage` of this line",
                    "fp64_arith_wave_instructions_per_stage": add + mul + fma,
                    "valu_wave_instructions_per_stage": valu,
                    "live_lanes_per_valu_instruction": lanes,
                    "fp64_flops_per_stage": flops, "achieved_TFLOPs": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / FP64_VALU_PEAK_TFLOPS,
                    # the time the stage would take if nothing but its fp64 add / mul / fma were issued, at full rate, and
                    # the same for every VALU instruction it executes (compares, selects, moves, DPP, address arithmetic)
                    "fp64_issue_floor_ms": {"at_2.4_GHz_nominal": (add + mul + fma) * FP64_ISSUE_CYCLES / N_SIMD / 2.4e9 * 1e3,
                                            "at_2.0_GHz_measured_under_load": (add + mul + fma) * FP64_ISSUE_CYCLES / N_SIMD / 2.0e9 * 1e3},
                    "all_valu_issue_floor_ms_at_2.0_GHz": valu * FP64_ISSUE_CYCLES / N_SIMD / 2.0e9 * 1e3,
                    "stage_ms": stage_ms,
                    "frac_of_hbm_roofline_at_the_fp64_issue_floor": 288.0 * 16777216 / ((add + mul + fma) * FP64_ISSUE_CYCLES / N_SIMD / 2.0e9) / 1e9 / HBM_PEAK_GBS,
                    "source": "profiles/%s_pmc_instruction_mix.json%s -- a COMMITTED counter profile of the stage benchmark, not this run; "
                              "flops at full width (masked lanes counted), fma = 2" % (rnd, " @ " + commit if commit else "")}
        except Exception:
            continue
    return None


def _profile_commit(path):
    """the commit that last touched a committed profile (None outside a git checkout, e.g. on the GPU box)"""
    try:
        import subprocess
        r = subprocess.run(["git", "log", "-1", "--format=%h", "--", path], cwd=ROOT, capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None
    except Exception:
        return None


def measured_traffic(workload):
    """HBM bytes per fused-stage launch.  A live run cannot profile itself, so this is read from the newest
    COMMITTED rocprofv3 PMC summary of this same command (profiles/rNN_hbm_traffic.json, made by
    profiles/pmc_traffic.py from FETCH_SIZE / WRITE_SIZE collected in separate passes); the source string names
    the file.  Returns (GB, source) or (None, None)."""
    if workload == "hydro_plm_hllc_rk2_256":
        # the single-march stage: one launch per stage, the two stages of a cycle (without / with FillDerived + dt) averaged
        path = os.path.join(ROOT, "profiles", "r06_hbm_traffic_hydro.json")
        try:
            with open(path) as f:
                k = json.load(f)["kernels"]
            commit = _profile_commit(path)
            stages = ("fused_s3_kernel<1, 2, 4, 0, 1>", "fused_s3_kernel<1, 2, 4, 2, 2>")
            return (sum(k[name]["hbm_total_GB"] for name in stages) / len(stages),
                    "profiles/r06_hbm_traffic_hydro.json%s -- a COMMITTED profile of this command, not this run (rocprofv3 --pmc "
                    "FETCH_SIZE / WRITE_SIZE in separate passes, reads calibrated on the dt kernel's known bytes; mean of the two "
                    "stages of a cycle)" % (" @ " + commit if commit else ""))
        except Exception:
            return None, None
    if workload != "mhd_ppm_hlld_vl2_256":
        return None, None
    # round 2 / 3: the two-kernel stage (x3 sweep + finishing x1/x2 march); round 1: three sweeps
    for fname, stage, note in (
            ("r06_hbm_traffic.json", ("fused_march_kernel<2, 3, 5, 3, false, 0, false>", "fused_m12f_kernel<2, 3, 5, 2, 1, false, false>"),
             "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, reads calibrated on the dt kernel's known bytes"),
            ("r05_hbm_traffic.json", ("fused_march_kernel<2, 3, 5, 3, false, 0, false>", "fused_m12f_kernel<2, 3, 5, 2, true, false>"),
             "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, reads calibrated on the dt kernel's known bytes"),
            ("r04_hbm_traffic.json", ("fused_march_kernel<2, 3, 5, 3, false, 0, false>", "fused_m12f_kernel<2, 3, 5, 2, true, false>"),
             "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, reads calibrated on the dt kernel's known bytes"),
            ("r03_hbm_traffic.json", ("fused_march_kernel<2, 3, 5, 3, false, 0>", "fused_m12f_kernel<2, 3, 5, 2>"),
             "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, reads calibrated on the dt kernel's known bytes"),
            ("r02_hbm_traffic.json", ("fused_march_kernel<2, 3, 5, 3, false, 0>", "fused_m12f_kernel<2, 3, 5, 2>"),
             "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, reads calibrated on the dt kernel's known bytes"),
            ("r01_hbm_traffic.json", ("fused_x1_kernel<2, 3, 5, false>", "fused_march_kernel<2, 3, 5, 2, false, 0>",
                                      "fused_march_kernel<2, 3, 5, 3, true, 2>"),
             "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, reads calibrated x1.60; THREE-SWEEP schedule of round 1")):
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                k = json.load(f)["kernels"]
            commit = _profile_commit(path)
            return (sum(k[name]["hbm_total_GB"] for name in stage),  # (KeyError -> the next older profile)
                    "profiles/%s%s -- a COMMITTED profile of this command, not this run (%s)"
                    % (fname, " @ " + commit if commit else "", note))
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--regions", type=int, default=3, help="timed regions of --steps cycles each; the median one is reported")
    ap.add_argument("--sustained", type=int, default=500, help="cycles of the 'sustained' figure of the N = 1 line (0: skip)")
    ap.add_argument("--workload", default="mhd_ppm_hlld_vl2_256", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-copies-base", action="store_true",
                    help="skip the second N = 1 figure (same steps with the same-rank ghost copies of an N > 1 run)")
    ap.add_argument("--unfused", action="store_true", help="use the flux-array path (for A/B)")
    ap.add_argument("--brick", type=int, default=0, help="cells per GPU and direction instead of the workload's 256 (tests)")
    ap.add_argument("--meshblock", type=int, default=0, help="meshblock size instead of the workload's 128 (tests)")
    ap.add_argument("--no-rehearsal", action="store_true",
                    help="skip 'rehearsal_8gpu_rank' (the N = 1 brick with its outer faces exchanged like those between bricks; ~10 s)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the 'other_workloads' figures of the default N = 1 run (BASELINE configs 2, 4's scheme, 5's mesh; ~20 s)")
    ap.add_argument("--amr-extra", action="store_true",
                    help="append the adaptive-mesh figure (BASELINE config 5 shape) as 'amr_blast_cfg5'; off by default so "
                         "that the kernels of the default command are those of the headline workload only")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    import torch
    import torch.distributed as dist
    from athenapk_amd import decks, driver

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # development aid (one-GPU boxes): APK_SHARE_GPU=1 puts every rank on cuda:0 and
    # APK_DIST_BACKEND=gloo stages the halo messages through the host; the driver's runs use
    # neither (one rank per GPU, RCCL)
    backend = os.environ.get("APK_DIST_BACKEND", "nccl")
    if os.environ.get("APK_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    deck, fluid, integrator, recon, riemann, brick, mb, desc = WORKLOADS[args.workload]
    if args.brick or args.meshblock:  # (a reduced copy of the workload for the launch tests: NOT the benchmark)
        brick, mb = args.brick or brick, args.meshblock or mb
        desc += " -- REDUCED to %d^3 per GPU in %d^3 meshblocks" % (brick, mb)
    if world not in RANK_GRID:
        raise SystemExit("supported GPU counts: %s" % sorted(RANK_GRID))
    grid = RANK_GRID[world]
    ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick * grid[d]) for d in range(3)]
    ov += ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
    ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann]
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def start(comm):
        """the sim initialised and warmed up; (sim, error) -- an error leaves no sim behind"""
        sim = None
        try:
            sim = driver.Simulation(decks.load(deck), ov, rank=rank, nranks=world, strict=False, comm=comm)
            if args.unfused:
                sim.set_fused(False)
            if os.environ.get("APK_OVERLAP") == "0":  # A/B switch: exchange synchronously
                sim.set_overlap(False)
            sim.initialize()
            for _ in range(args.warmup):
                sim.step()
            torch.cuda.synchronize()
            if comm is None and os.environ.get("APK_BENCH_INJECT_START_FAILURE") == str(rank):  # (test hook)
                raise RuntimeError("injected start-up failure on rank %d" % rank)
            return sim, None
        except Exception as e:  # noqa: BLE001 -- whatever it is, the other ranks must hear about it
            try:
                if sim is not None:
                    sim.close()
            except Exception:
                pass
            return None, e

    sim, err = start(None)
    if world > 1:
        # The native RCCL transport has its first multi-GPU exchange HERE (one-GPU boxes can only self-test it): if
        # it fails on ANY rank during initialisation or warm-up, every rank starts over on the torch.distributed
        # callback transport -- loudly, and the line says which transport carried the timed steps.
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            print("[bench] rank %d: start-up failed on some rank (%r here); retrying with comm='torch'" % (rank, err),
                  file=sys.stderr, flush=True)
            if sim is not None:
                sim.close()
            sim, err = start("torch")
    if err is not None:
        raise err
    info = sim.info
    # THREE timed regions of exactly --steps cycles each, every one bracketed by barrier + synchronize and reduced with
    # MAX over the ranks; `value` / `ms_per_step` are those of the MEDIAN region, all three are listed in
    # `timed_regions_ms`.  No event records in these loops (round-5 review): the kernels' HIP-event durations behind
    # `roofline` come from a FOURTH region of the same --steps cycles right after them, with the in-library timing on
    # (two event records per launch on the stream the kernels run on).
    regions = []
    comm_before = sim.comm_stats() if world > 1 else None
    for _ in range(max(1, args.regions)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sim.step()
        torch.cuda.synchronize()
        barrier()
        e = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        regions.append(e)
    med = sorted(range(len(regions)), key=lambda q: regions[q])[len(regions) // 2]
    elapsed = regions[med]
    comm_after = sim.comm_stats() if world > 1 else None
    sim.kernel_timing(True)
    sim.read_kernel_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.step()
    torch.cuda.synchronize()
    barrier()
    event_region = time.perf_counter() - t0
    timing = sim.read_kernel_timing()
    sim.kernel_timing(False)
    comm_stats = comm_after
    x1_direct_per_cycle = sim.x1_direct_exchanges() / max(1, sim.ncycle)
    skipped_per_cycle = sim.skipped_local_exchanges() / max(1, sim.ncycle)
    overlapped_per_cycle = sim.overlapped_exchanges / max(1, sim.ncycle)
    # 500 cycles back to back (N = 1): does the rate hold once the clocks have settled under the power cap?
    sustained = None
    if world == 1 and not args.unfused and args.sustained > 0 and not (args.brick or args.meshblock):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.sustained):
            sim.step()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        sustained = {"value": int(info.zones_total) * args.sustained / e2, "unit": "cell-updates/s", "cycles": args.sustained,
                     "ms_per_step": e2 / args.sustained * 1e3}
    # The honest base of a weak-scaling curve: at N = 1 every block face is a same-rank face and direct
    # neighbour addressing skips ALL ghost copies, while at N > 1 the faces between bricks are packed, sent
    # and unpacked.  Same workload once more with the same-rank ghost copies done as at N > 1 (a second
    # figure; `value` stays the default configuration's).
    copies_base = None
    if world == 1 and not args.unfused and not args.no_copies_base and sim.skipped_local_exchanges() > 0:
        sim.set_direct_neighbors(False)
        for _ in range(max(1, args.warmup)):
            sim.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            sim.step()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        copies_base = {"value": int(info.zones_total) * args.steps / e2, "unit": "cell-updates/s", "ms_per_step": e2 / args.steps * 1e3,
                       "what": "the same %d steps with APK_DIRECT_NEIGHBORS=0: same-rank ghost zones copied (+ ConsToPrim) after every "
                               "stage as they are between the bricks of an N > 1 run" % args.steps}

    if rank == 0:
        zones_total = int(info.zones_total)
        zones_local = int(info.zones_local)
        value = zones_total * args.steps / elapsed
        nstages = len(GAM0[integrator])
        per_kernel = {k: (ms / n if n else 0.0) for k, (ms, n) in timing.items()}
        # with the exchange overlapped the x1 sweep of a stage is several launches (main window +
        # slabs): account it per stage (= per finishing-sweep launch)
        fin = "fused_x3" if info.ndim == 3 else "fused_x2"
        if timing[fin][1]:
            per_kernel["fused_x1"] = timing["fused_x1"][0] / timing[fin][1]
        # one "launch" of the fused stage = its three sweep kernels back to back
        if args.unfused:
            stage_ms = per_kernel["fluxes"] + per_kernel["update"] + per_kernel["dedner"]
            stage_name = "flux-array stage: CalculateFluxes + UpdateWithFluxDivergence + DednerSource"
        else:
            # the high-order (PPM+HLLD) stage; the VL2 donor-cell predictor is timed in its own slots
            stage_ms = per_kernel["fused_x1"] + per_kernel["fused_x2"] + per_kernel["fused_x3"]
            two_kernel = info.ndim == 3 and per_kernel["fused_x2"] == 0.0 and per_kernel["fused_x3"] > 0.0
            if two_kernel:
                stage_name = ("fused %s+%s stage, two kernels: x3 sweep writing its flux difference (timing slot fused_x3) "
                              "+ ONE march doing x1 (DPP stencil) and x2 (LDS ring) that finishes the stage: RK update%s, "
                              "ConsToPrim of the interior, dt in the last stage (slot fused_x1)"
                              % (recon.upper(), riemann.upper(), ", Dedner" if fluid == "glmmhd" else ""))
            else:
                stage_name = ("fused %s+%s stage: x1 DPP sweep + x2 march + x3 march (+RK update%s, +ConsToPrim of "
                              "the interior, +dt in the last stage)" % (recon.upper(), riemann.upper(),
                                                                        " +Dedner" if fluid == "glmmhd" else ""))
        # high-order stages only (for vl2: the corrector, gam0 = 0)
        ho = [g0 for n, g0 in enumerate(GAM0[integrator]) if not (integrator == "vl2" and n == 0)]
        b_stage = sum(B_STAGE[fluid][0 if g0 == 0.0 else 1] for g0 in ho) / len(ho)
        achieved = b_stage * zones_local / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
        b_cycle = sum(B_STAGE[fluid][0 if g0 == 0.0 else 1] for g0 in GAM0[integrator]) + nstages * B_C2P[fluid]
        traffic_gb, traffic_src = (None, None) if args.unfused else measured_traffic(args.workload)
        dominant = max(("fused_x1", "fused_x2", "fused_x3", "fused_dc_x1", "fused_dc_x2", "fused_dc_x3", "fluxes",
                        "cons_to_prim", "copy_regions", "update", "min_dt"),
                       key=lambda k: timing[k][0])
        out = {
            "metric": "cell-updates/s (zone-cycles/s) for 3D %s, uniform grid" % (
                "MHD PPM+HLLD" if fluid == "glmmhd" else "hydro %s+%s" % (recon.upper(), riemann.upper())),
            "value": value,
            "unit": "cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "timed_regions_ms": [e * 1e3 for e in regions],
            "timed_region_reported": "median of %d regions of %d cycles each (barrier + synchronize around each, max over ranks; no event records inside)" % (len(regions), args.steps),
            "kernel_event_region_ms": event_region * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "description": desc,
                       "mesh": [int(info.nx[0]), int(info.nx[1]), int(info.nx[2])],
                       "meshblock": [mb, mb, mb], "blocks_per_gpu": int(info.nblocks_local),
                       "integrator": integrator, "nstages": nstages, "nghost": int(info.ng),
                       "path": "flux-array" if args.unfused else "fused",
                       "parallelism": "domain decomposition, %dx%dx%d GPU grid" % grid,
                       "comm_backend": (("rccl, native transport of the C++ host (ncclSend/ncclRecv groups on a halo stream)"
                                         if sim.comm_kind == "rccl" else "torch.distributed callbacks over " + backend)
                                        if world > 1 else None),
                       "overlapped_exchanges_per_cycle": overlapped_per_cycle if world > 1 else None,
                       # collectives per cycle on the reduction communicator (dt and the c_h estimate travel together)
                       "reductions_per_cycle": ((comm_stats["reductions"] - comm_before["reductions"]) / (args.steps * len(regions))) if comm_stats else None,
                       "halo_exchanges_per_cycle": ((comm_stats["exchanges"] - comm_before["exchanges"]) / (args.steps * len(regions))) if comm_stats else None,
                       # stage boundaries per cycle whose same-rank ghost copies were skipped (direct neighbour addressing)
                       "same_rank_ghost_copies_skipped_per_cycle": skipped_per_cycle,
                       # exchanges per cycle whose x1 strips bypassed the pack / unpack kernels (apk_stage_args.x1_halo)
                       "exchanges_with_x1_strips_in_the_buffers_per_cycle": x1_direct_per_cycle if world > 1 else None},
            "cell_stage_updates_per_s": value * nstages,
            "roofline": {
                # what binds the stage kernels (SQ counters); `achieved` / `peak` / `frac` are priced against HBM bandwidth,
                # as north_star asks -- `priced_against` -- and against the fp64 vector roof in `valu`
                # (the hydro single march as well: vector issue busy 94 % of its run time, profiles/r06_pmc_sq_hydro.json)
                "bound": "hbm" if args.unfused else "fp64_valu_issue",
                "priced_against": "hbm",
                "kernel": stage_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic_gb,
                "traffic_unit": "GB per fused-stage launch (algorithmic: %.2f GB)" % (b_stage * zones_local / 1e9),
                "traffic_source": traffic_src,
                "traffic_note": ("of the 8.85 GB: 2.4 GB are the x3 sweep's flux differences written and read back (the two-kernel stage's "
                                 "own 144 B per cell on top of the 216 algorithmic ones), the line-aligned rows of round 6 fetch nine lines "
                                 "per 128-cell row where natural rows average 8.4 (+0.9 GB against profiles/r05_hbm_traffic.json, for "
                                 "-1.3 % time: profiles/r06_row_pitch_ab.txt); the kernels are bound by vector issue, not by these bytes")
                                if (args.workload == "mhd_ppm_hlld_vl2_256" and traffic_gb is not None) else None,
                "algorithmic_bytes_per_cell_stage": b_stage,
                "cells_per_launch": zones_local,
                "stage_ms": stage_ms,
                "dc_predictor_stage_ms": per_kernel["fused_dc_x1"] + per_kernel["fused_dc_x2"] + per_kernel["fused_dc_x3"],
                "frac_of_measured_copy_bandwidth": achieved / HBM_COPY_GBS,
                "binding_limit": "fp64 vector-ALU issue (SQ counters: profiles/%s) "
                                 % ("r06_pmc_sq_hydro.json, r06_pmc_instruction_mix_hydro.json" if fluid != "glmmhd" else
                                    ("r06_pmc_sq_wenoz.json" if "wenoz" in args.workload else "r06_pmc_sq.json, r06_pmc_instruction_mix.json")) +
                                 "under the 1400 W power cap (effective clock profiles/r06_clock.json), next to the access pattern of "
                                 "the march (profiles/r04_ubench_march_traffic.jsonl); see roofline.valu for the compute roof",
                "note": "`peak` is the 8 TB/s HBM3E spec, frac_of_measured_copy_bandwidth prices against the 6.29 TB/s a float4 copy "
                        "reaches (MI355X_MICROARCH.md); kernel resources (VGPRs, scalar spills, LDS): profiles/r06_kernel_resources.txt. "
                        "PRODUCT build (FMA contraction, rsq/rcp-based roots and reciprocals): L1 norms within 1e-12 of the bit-exact "
                        "parity build (which is what the parity tests pin); PPM states can differ by ~1e-8 after a few cycles "
                        "(a last-bit difference flips an extremum test); DESIGN.md sections 4 and 7",
                "per_kernel_avg_ms": per_kernel,
                "dominant_kernel": ROCPROF_KERNEL.get(dominant, dominant) if fluid == "glmmhd" and recon == "ppm" else dominant,
                "dominant_timing_slot": dominant,
                "whole_cycle": {"algorithmic_bytes_per_zone_cycle": b_cycle,
                                "achieved": value / world * b_cycle / 1e9,
                                "frac": value / world * b_cycle / 1e9 / HBM_PEAK_GBS,
                                "note": "SURVEY 8(d)'s per-zone-cycle bytes INCLUDING the ConsToPrim stores and re-reads of "
                                        "the reference's cycle, which this cycle no longer performs where it can derive the "
                                        "primitives in registers: a figure of how much reference traffic per second the cycle "
                                        "stands for, NOT the north-star roofline fraction (that is `frac` / `general_stage.frac`)"},
            },
        }
        if sustained:
            out["sustained_%d_cycles" % sustained["cycles"]] = sustained
        if copies_base:
            out["weak_scaling_base_with_ghost_copies"] = copies_base
        if world == 1 and fluid == "glmmhd" and not args.unfused:
            try:
                sim.close()
                out["roofline"]["general_stage"] = general_stage_bench(recon, riemann)
                if recon == "ppm" and riemann == "hlld":
                    out["roofline"]["valu"] = valu_roofline(out["roofline"]["general_stage"]["ms_per_stage"])
            except Exception as e:  # supplementary figure; never lose the headline
                out["roofline"]["general_stage"] = {"error": repr(e)}
        if world == 1 and not args.unfused and not args.no_rehearsal and not (args.brick or args.meshblock):
            try:
                out["rehearsal_8gpu_rank"] = rehearsal_8gpu_rank(args.workload, elapsed / args.steps * 1e3)
            except Exception as e:  # supplementary figure
                out["rehearsal_8gpu_rank"] = {"error": repr(e)}
        if world == 1 and not args.unfused and not args.no_other_workloads and not (args.brick or args.meshblock) \
                and args.workload == "mhd_ppm_hlld_vl2_256":
            out["other_workloads"] = other_workloads()
        if world == 1 and args.amr_extra:
            try:
                out["amr_blast_cfg5"] = amr_blast_bench()
            except Exception as e:  # supplementary figure
                out["amr_blast_cfg5"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(fluid, integrator, recon, riemann)
            except Exception as e:  # the baseline is informational; never lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "cell-updates/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
