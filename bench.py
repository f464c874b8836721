#!/usr/bin/env python
"""bench.py -- transverse slices/s of the per-zeta-slice hot path on MI355X.

One "step" = one transverse slice (deposit -> Psi/Ez/Bz Poisson solves -> explicit deposit ->
Bx/By multigrid -> gather+push) of the BASELINE.md section 3 synthetic deck: 1024 x 1024 cells,
2x2 = 4 plasma particles per cell, order-2 shapes, Gaussian fixed-ppc driver, explicit solver,
hipace.dt = 0.  The default run times one whole box (1024 slices = one time step, including the
per-step plasma re-initialisation); slices/s = slices / wall time, particle-pushes/s = 4*1024^2
times that.  All inputs are generated on the device before the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  The path shards as the
reference does, over time steps (rank r runs steps r, r+N, ...: Hipace.cpp:400-401), each rank
sweeping the whole box; weak scaling, value = total slices of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(n, ppc2):
    """SURVEY 8(d) algorithmic traffic per slice of each kernel (fp64), C = n^2 cells, P = ppc2*C."""
    C = n * n
    P = ppc2 * C
    return {
        "deposit_current": 56 * P + 4 * 8 * C,
        "explicit_deposit": 56 * P + 4 * 8 * C + 2 * 8 * C,
        "advance_plasma": (40 + 8) * P + 80 * P + 5 * 8 * C,
        "poisson": 3 * 4 * 16 * C,
    }


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of `kernel_prefix` from the committed rocprofv3 --pmc summary (two separate
    passes, FETCH_SIZE and WRITE_SIZE, scripts/pmc_traffic.py).  Units are KB; on gfx950 FETCH_SIZE reports
    half of the bytes of a coalesced read (MI355X_MICROARCH.md, HBM section) -- calibrated here on
    k_copy_comps / k_zero_comps / k_init_plasma, whose byte counts are known: FETCH x2, WRITE x1.
    Only valid for the default 1024^2 x 4 ppc workload the counters were collected on."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01h_pmc_fetch_write_per_kernel.csv")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["kernel"].startswith(kernel_prefix):
                return (2.0 * float(r["FETCH_SIZE_raw_per_launch"]) + float(r["WRITE_SIZE_raw_per_launch"])) * 1024.0
    return None


def cpu_baseline(n, ppc, nslices):
    """Time the CPU oracle (single-thread restatement of the reference's serial path) on the head
    `nslices` slices of the same deck."""
    from hipace_amd import decks
    from oracle import oracle as O
    deck = decks.synthetic(n, 1024, ppc)
    eng = O.Engine(deck)
    eng.begin_step()
    t0 = time.perf_counter()
    for k in range(nslices):
        eng.solve_slice(deck["nz"] - 1 - k)
    dt = time.perf_counter() - t0
    return dict(value=nslices / dt, unit="slices/s", cores=1, kind="port",
                sample=f"oracle (serial C++ restatement, g++ -O2), head {nslices} slices of the same "
                       f"{n}x{n}x1024 {ppc * ppc}ppc deck, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024, help="timed slices")
    ap.add_argument("--warmup", type=int, default=64, help="untimed warm-up slices")
    ap.add_argument("--n", type=int, default=1024, help="transverse cells per side")
    ap.add_argument("--ppc", type=int, default=2, help="plasma particles per cell per direction")
    ap.add_argument("--tile", type=int, default=16, help="particle tile size (0, 16, 32)")
    ap.add_argument("--sort-period", type=int, default=128, help="max slices between particle re-sorts (adaptive below)")
    ap.add_argument("--cpu-slices", type=int, default=4, help="slices of the CPU baseline sample (0 = skip)")
    ap.add_argument("--profile-stride", type=int, default=7,
                    help="HIP-event phase timers (and with them the roofline's kernel duration) on every n-th slice of the "
                         "timed region: the 11 event records of a timed slice cost 4.5 %% of it")
    ap.add_argument("--inflight", type=int, default=1,
                    help="time steps in flight on one GPU (hipace_amd/pipeline.py::run_local_pipeline): L engines on L "
                         "streams, step s+1 trails step s by the per-slice beam hand-off.  Needs --steps >= L boxes.  "
                         "Default 1; with --gpus N > 1 every rank runs L stages of the ring (gloo-tested, not yet on RCCL)")
    ap.add_argument("--laser-solver", choices=["fft", "multigrid"], default="fft",
                    help="--config5: lasers.solver_type (multigrid = hpmg system type 2, the reference's default)")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE config 5 without its ionisation (not built): laser_blowout_wake 1024x1024x2048, 4 ppc, a "
                         "Gaussian laser pulse drives the wake and is advanced by the FFT envelope solver on every slice; "
                         "the three time levels of the envelope (3 x 34 GB) stay in HBM (not the judged bench line)")
    ap.add_argument("--config2", action="store_true",
                    help="BASELINE config 2 instead of the headline workload: linear_wake 256x256x512, 4 ppc, "
                         "predictor-corrector Bx/By solver (not the judged bench line)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hipace_amd import api, decks

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    nz = 1024
    deck = decks.synthetic(args.n, nz, args.ppc)
    if args.config2:
        nz, args.n, args.ppc = 512, 256, 2
        deck = decks.predictor_corrector(decks.linear_wake(), 4.0e-2, 30, 0.05)
        deck.update(nx=256, ny=256, nz=nz, plasma_ppc=(2, 2))
        args.steps = min(args.steps, nz) if args.inflight <= 1 else args.steps
        # (the loop's cost depends on the slice: every run_slices() starts a box from its head)
        args.cpu_slices = 0
    if args.config5:
        nz = 2048
        deck = decks.synthetic(args.n, nz, args.ppc)
        deck.update(beam_profile=-1, lo=(-20.0, -20.0, -15.0), hi=(20.0, 20.0, 6.0), laser_on=1, laser_a0=4.5, laser_w0=4.0,
                    laser_L0=2.0, laser_lambda0=0.08, laser_solver=2 if args.laser_solver == "multigrid" else 1, dt=5.0)
        args.cpu_slices = 0
        args.inflight = 1
    eng = api.SliceEngine(deck, device=local, tile_size=args.tile, sort_period=args.sort_period)
    lanes = 1
    if args.inflight > 1:
        lanes = max(1, min(args.inflight, args.steps // nz))      # whole boxes only: head slices are cheaper than the rest
    engines = [eng] + [api.SliceEngine(deck, device=local, tile_size=args.tile, sort_period=args.sort_period)
                       for _ in range(lanes - 1)]

    def run_slices(count, profile=False):
        done = 0
        while done < count:
            eng.begin_step()
            m = min(nz, count - done)
            for k in range(m):
                eng.solve_slice(nz - 1 - k)
            done += m

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    run_slices(args.warmup)
    groups = None
    staged = lanes > 1 or (args.config5 and world > 1)       # the laser's time levels travel through the multi-stage ring only
    if staged:
        from hipace_amd.pipeline import make_edge_groups, run_local_pipeline
        groups = make_edge_groups(world)
    if lanes > 1:
        run_local_pipeline(engines, lanes, torch.device("cuda", local), slices_per_step=max(2, args.warmup))   # warm every lane
    for e in engines:
        e.set_profiling(True, stride=args.profile_stride)
    barrier()
    t0 = time.perf_counter()
    if staged:
        # `lanes` pipeline stages per GPU; every stage sweeps whole boxes.  world > 1 (opt-in, --inflight): stage
        # r*lanes + l on rank r, RCCL only on the rank-to-rank edges
        boxes = max(1, args.steps // nz) * world
        args.steps = run_local_pipeline(engines, boxes, torch.device("cuda", local), rank=rank, world=world, groups=groups)
    elif world == 1:
        run_slices(args.steps)
    else:
        # ring pipeline over time steps: every rank sweeps `steps` slices of its own step(s); the beam
        # slices travel rank -> rank+1 through RCCL (hipace_amd/pipeline.py)
        from hipace_amd.pipeline import run_pipeline
        per_step = max(min(args.steps, nz), 2 * world)      # the ring needs 2 slices of skew per rank
        steps_per_rank = max(1, args.steps // nz)
        solved = run_pipeline(eng, rank, world, world * steps_per_rank, torch.device("cuda", local),
                              slices_per_step=per_step)
        args.steps = solved
    barrier()
    dt = time.perf_counter() - t0
    phases, nprof = eng.phase_times()
    eng.set_profiling(False)
    for e in engines[1:]:            # phase times: mean over the lanes (intervals overlap in wall time)
        ph, n = e.phase_times()
        e.set_profiling(False)
        for k in phases:
            phases[k] += ph[k]
        nprof += n
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        total = args.steps * world
        ab = algorithmic_bytes(args.n, args.ppc * args.ppc)
        per_kernel = {k: phases[k] / max(nprof, 1) for k in phases}
        dom = "deposit_current"
        achieved = ab[dom] / (per_kernel[dom] * 1e-3) / 1e9 if per_kernel[dom] > 0 else 0.0
        out = {
            "metric": "transverse slices/s at 256^2 x 4ppc (predictor-corrector solver)" if args.config2 else
                      f"transverse slices/s at 1024^2 x 4ppc with a laser envelope (explicit solver, {args.laser_solver} envelope solver)" if args.config5 else
                      "transverse slices/s at 1024^2 x 4ppc (explicit solver)",
            "value": total / dt, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "particle_pushes_per_s": total / dt * args.ppc * args.ppc * args.n * args.n,
            "config": {"workload": ("linear_wake.normalized 256x256x512, 4 ppc, order 2, predictor-corrector Bx/By solver "
                                    "(tolerance 4e-2, <= 30 iterations, mixing 0.05), dt=0 (BASELINE.json configs[1])") if args.config2 else
                                   (f"laser_blowout_wake {args.n}x{args.n}x{nz}, 4 ppc, order 2, Gaussian laser a0=4.5 + {args.laser_solver} envelope "
                                    "solver every slice, no ionisation (BASELINE.json configs[4] without its ionisation)") if args.config5 else
                                   (f"blowout_wake synthetic {args.n}x{args.n}x{nz}, {args.ppc * args.ppc} ppc, "
                                    "order 2, explicit Bx/By solver, dt=0 (BASELINE.md section 3)"),
                       "parallelism": f"time-step pipeline x{world}" + (f", {lanes} steps in flight per GPU" if lanes > 1 else "")},
            "steps_in_flight": lanes,
            "phase_ms_per_slice": per_kernel,
            "vcycles_per_slice": eng.stats()["vcycles"] / max(eng.stats()["slices"], 1),
            "laser_vcycles_per_slice": (eng.laser_vcycles() / max(eng.stats()["slices"], 1)) if args.config5 else None,
            "pc_iterations_per_slice": eng.pc_stats()[0] / max(eng.stats()["slices"], 1) if args.config2 else None,
            "particle_sorts": eng.sorts() if args.tile else 0,
            "halo_fallbacks": eng.fallbacks() if args.tile else 0,
            "roofline": {"bound": "hbm", "kernel": "k_deposit_tiled<2,%d>" % args.tile if args.tile else "k_deposit_current<2>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("void hps::k_deposit_tiled<2, 16, 51>") if (args.tile == 16 and args.n == 1024 and args.ppc == 2 and not args.config5) else None,
                         "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": per_kernel[dom]},
        }
        if args.cpu_slices > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n, args.ppc, args.cpu_slices)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
