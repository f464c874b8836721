#!/usr/bin/env python
"""bench.py -- transverse slices/s of the per-zeta-slice hot path on MI355X.

One "step" = one transverse slice (deposit -> Psi/Ez/Bz Poisson solves -> explicit deposit ->
Bx/By multigrid -> gather+push) of the BASELINE.md section 3 synthetic deck: 1024 x 1024 cells,
2x2 = 4 plasma particles per cell, order-2 shapes, Gaussian fixed-ppc driver, explicit solver,
hipace.dt = 0.  The default run times one whole box (1024 slices = one time step, including the
per-step plasma re-initialisation); slices/s = slices / wall time, particle-pushes/s = 4*1024^2
times that.  All inputs are generated on the device before the timed region.

--steps K < 1024: the cost of a slice varies along the box (the head slices ahead of the driver see an
unperturbed sheet: fewer multigrid V-cycles, even tiles), so a short run is NOT taken at the head: the
engine first runs (untimed) down to slice --start-slice, where a window of K slices costs what the whole
box costs on average, then W warm-up slices, then exactly K timed slices.  The line says which slices were
timed and how many V-cycles they needed.

Several time steps in flight: after the headline measurement (one engine, `value`) the same workload is timed with
--inflight L (default 3) engines on L streams of the GPU, L consecutive time steps trailing one another by the per-slice
beam hand-off (hipace_amd/pipeline.py::run_lanes; the reference gets the same effect with several MPI ranks per device,
Hipace.cpp:400-401): `value_steps_in_flight`, `steps_in_flight`.  A slice is a chain of dependent, often latency-bound
kernels; the only independent work there is are the other time steps of the pipeline.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL; `python bench.py --gpus N` without
torchrun re-executes itself under `python -m torch.distributed.run`).  The path shards as the reference
does, over time steps (rank r runs steps r, r+N, ...: Hipace.cpp:400-401), each rank sweeping the whole box
and handing every slice's beam block to the next rank over the C-ABI RCCL ring (hps_ring_*, one message per
slice, event-ordered, no host synchronisation per slice).  Weak scaling, value = total slices of all
ranks / max-over-ranks time.  K >= 1024: whole steps, pipeline fill included.  K < 1024: the pipeline is
filled before the clock starts (rank r runs 2r slices behind rank 0, as the hand-off requires), then every
rank times K slices.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

for _i, _a in enumerate(sys.argv):          # --edge ipc|rccl: the kind of edge between processes (read by the library when the ids are made)
    if _a == "--edge" and _i + 1 < len(sys.argv):
        os.environ["HPS_RING_EDGE"] = sys.argv[_i + 1]
    elif _a.startswith("--edge="):
        os.environ["HPS_RING_EDGE"] = _a.split("=", 1)[1]
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("HPS_RING_TIMEOUT_S", "300")      # a ring that stops making progress fails loudly well inside the driver's limit
if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "--same-device" not in sys.argv:
    # For RCCL edges -- HPS_RING_EDGE=rccl, or the fall-back of a ring whose ipc probe message fails (RingTransport).  Before HIP starts: the engine's stream, the ring's send and receive streams and torch's own streams must not end up
    # sharing a hardware queue (a send queued behind a receive that waits for its data would close a circle around the ring)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if "--config5" not in sys.argv:
        # A receive posted ahead is an RCCL kernel that waits on the device for its data: by default 16 workgroups for a
        # 1 MB beam block (measured: grid 4096 x 256 threads), i.e. 16 CUs' worth of slots held while the engine runs.  Two
        # channels carry a block in 84 us (12 GB/s against the 1.3 GB/s a slice hand-off needs); the laser's 33 MB per slice
        # (config 5) keep RCCL's default.
        os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "2")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CPU_THREADS_DEFAULT = 16  # OpenMP leg of the CPU baseline: fastest on the 256-core host of the GPU box, 9.5x the serial leg; 64 threads
                          # are already slower and 256 slower than one (profiles/r02b_cpu_threads.json)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_SUMMARY = "r06_pmc_fetch_write_per_kernel.csv"       # profiles/: rocprofv3 --pmc summary of the shipped kernels
PMC_FP64 = "r06_pmc_fp64_per_kernel.csv"                  # profiles/: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per launch (scripts/collect_flops.sh)
FP64_VALU_PEAK_TFLOPS = 78.6  # vector fp64: half of MI355X_MICROARCH.md's 157.3 TFLOP/s fp32 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop
                              # x 2.4 GHz, every instruction an FMA); scripts/ubench/mfma64.hip measured 71 on this part
START_SLICE_DEFAULT = 700  # short runs start here (from the head); see profiles/r02a_slice_cost_profile.json


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


def slice_bytes(n, ppc2, n_vcycles):
    """SURVEY 8(d), last rows of the table: algorithmic bytes of one whole slice (explicit solver, no laser) with the
    reference's pass structure -- deposition, explicit deposition, gather + push, three Poisson solves of 9 passes,
    right-hand sides / -grad Psi / Sx, Sy / AddRhoIons, the zero and shift slab operations, hpmg solve1 at the measured
    number of V-cycles -- and the survey's fused lower bound (ideal 4-pass Poisson solves, no zero / shift traffic, the push
    of slice k and the deposition of slice k-1 sharing the particle reads: 184 B per particle)."""
    C = n * n
    P = ppc2 * C
    mg = 17 * 8 * C + (4.0 / 3.0) * 23 * 8 * C * n_vcycles
    reference = (56 * P + 32 * C) + (56 * P + 48 * C) + (128 * P + 40 * C) + 3 * (9 * 16 * C + 8 * C) + 176 * C + 152 * C + mg
    fused = 184 * P + (32 + 48 + 40) * C + 3 * 4 * 16 * C + 176 * C + mg
    return reference, fused


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of `kernel_prefix` from the committed rocprofv3 --pmc summary (two separate
    passes, FETCH_SIZE and WRITE_SIZE, scripts/pmc_traffic.py).  Units are KB; on gfx950 FETCH_SIZE reports
    half of the bytes of a coalesced read (MI355X_MICROARCH.md, HBM section) -- calibrated on
    k_copy_comps / k_zero_comps / k_init_plasma, whose byte counts are known: FETCH x2, WRITE x1.
    Only valid for the default 1024^2 x 4 ppc workload the counters were collected on.  Returns
    (bytes, source) -- NOT measured in this run: the counters need their own rocprofv3 passes."""
    import csv
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)      # no older summary stands in for a missing one: traffic = null then
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["kernel"].startswith(kernel_prefix):
                b = (2.0 * float(r["FETCH_SIZE_raw_per_launch"]) + float(r["WRITE_SIZE_raw_per_launch"])) * 1024.0
                return b, f"committed rocprofv3 --pmc summary profiles/{PMC_SUMMARY} (FETCH_SIZE x2 + WRITE_SIZE, separate passes; not collected in this run)"
    return None, None


def pmc_slice_bytes():
    """HBM bytes per SLICE summed over every kernel of the committed --pmc summary (FETCH_SIZE x2 + WRITE_SIZE, as pmc_traffic):
    launches x bytes per launch of every kernel / the slices of that run (= the launches of the deposition kernel, one per
    slice).  Counter traffic of the whole schedule, re-sorts amortised as they fell in the counted window; None without a summary."""
    import csv
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    tot, nsl = 0.0, 0
    with open(path) as f:
        for r in csv.DictReader(f):
            try:
                b = (2.0 * float(r["FETCH_SIZE_raw_per_launch"]) + float(r["WRITE_SIZE_raw_per_launch"])) * 1024.0
            except ValueError:
                continue
            if b != b:      # nan: the kernel has no WRITE_SIZE row
                continue
            tot += b * int(r["launches"])
            if r["kernel"].startswith("void hps::k_deposit_tiled<2, 16, 51"):
                nsl = int(r["launches"])
    return tot / nsl if nsl else None


def pmc_fp64(kernel_prefix):
    """fp64 flops and VALU instructions per launch of `kernel_prefix` from the committed SQ instruction counters (wave-level
    instruction counts x 64 lanes: ADD + MUL + 2 FMA + TRANS; exec-masked lanes count as if active -- an upper bound of the
    useful flops).  None without the summary."""
    import csv
    path = os.path.join(ROOT, "profiles", PMC_FP64)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["kernel"].startswith(kernel_prefix):
                f64 = sum(float(r[k]) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
                return dict(flops=float(r["fp64_flops_per_launch"]), fp64_wave_instructions=f64, valu_wave_instructions=float(r["SQ_INSTS_VALU"]))
    return None


CONFIG5_ROWS = (   # row, what, kernel-name patterns (mangled names of the committed kernel trace), bytes per slice as f(C cells, Pe electrons, Pi ions) or None
    ("advance_plasma_electrons", "k_advance_tiled<LASER> of the electron sheet (the electrons the dopant has released ride in its tail)",
     ("k_advance_tiledILi2ELi16ELb1ELb0E",), lambda C, Pe, Pi: 128 * Pe + 6 * 8 * C),
    ("advance_plasma_ions_adk", "k_advance_tiled<LASER, IONIZE> of the N dopant: gather, ADK decision per macro-atom, push (tiles without field above threshold skip)",
     ("k_advance_tiledILi2ELi16ELb1ELb1E", "k_ion_field_bounds", "k_ionize"), lambda C, Pe, Pi: 136 * Pi + 6 * 8 * C),
    ("deposit_current_both_species", "k_deposit_tiled<LASER>, one launch per species and slice",
     ("k_deposit_tiled",), lambda C, Pe, Pi: 56 * (Pe + Pi) + 2 * 5 * 8 * C),
    ("explicit_deposit_both_species", "k_explicit_tiled<LASER>, one launch per species and slice",
     ("k_explicit_tiled",), lambda C, Pe, Pi: 56 * (Pe + Pi) + 2 * 7 * 8 * C),
    ("envelope_solve", "the slice's envelope advance on the engine's laser stream: right-hand side, phases, the solve (hpmg type 2: k2_*; fft: rocFFT's "
                       "fft_rtc / transpose_rtc kernels + k_laser_divide), store, |a|^2",
     ("k2_", "k_laser_", "fft_rtc", "transpose_rtc"), None),
    ("poisson", "3 solves: DST along x, tridiagonal solves along y, DST along x", ("k_dst_", "k_tridiag", "k_dense"), lambda C, Pe, Pi: 3 * (3 * 16 + 8) * C),
    ("mg_solve1", "hpmg solve1 of Bx, By", ("k_smooth", "k_lower_v", "k_post_norms", "k_hierarchy_gradpsi"), None),
    ("sort_slab_other", "re-sorts (both species), slab shift / zero, fills and copies", ("rocprim", "k_permute", "k_cell_keys", "k_shift_zero", "k_tile_", "k_run_starts",
                                                                                             "k_rank_keys", "fillBuffer", "copyBuffer"), None),
)


def config5_rows(si, solver, n, ppc2):
    """roofline.per_kernel of a config-5 run from the COMMITTED rocprofv3 --kernel-trace summary of the same command (profiles/
    r06_config5_<units>_<solver>_kernel_stats.csv: scripts/quick_prof.sh): kernel time per slice of every group of kernels -- the ion
    species' push with its ADK decisions and the envelope solve as rows of their own; the envelope's kernels run on a second stream
    beside the slice's chain, so the rows add up to more than a slice takes.  None without the summary."""
    import csv
    name = f"r06_config5_{'si' if si else 'norm'}_{'mg' if solver == 'multigrid' else 'fft'}_kernel_stats.csv"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path) or n != 1024:
        return None
    rows_csv = list(csv.DictReader(open(path)))
    nsl = next((int(r["Calls"]) for r in rows_csv if "k_laser_rhs" in r["Name"]), 0)
    if not nsl:
        return None
    C = n * n
    Pe, Pi = ppc2 * C, C
    out, used = {}, set()
    for key, what, pats, fb in CONFIG5_ROWS:
        ns = 0.0
        launches = 0
        for i, r in enumerate(rows_csv):
            if i not in used and any(p in r["Name"] for p in pats):
                used.add(i)
                ns += float(r["TotalDurationNs"])
                launches += int(r["Calls"])
        us = ns / nsl / 1e3
        b = fb(C, Pe, Pi) if fb else None
        out[key] = {"kernels": what, "us_per_slice": us, "launches_per_slice": launches / nsl, "algorithmic_bytes": b,
                    "achieved": b / (us * 1e-6) / 1e9 if (b and us > 0) else None,
                    "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if (b and us > 0) else None}
    rest = sum(float(r["TotalDurationNs"]) for i, r in enumerate(rows_csv) if i not in used and "k_laser_init" not in r["Name"] and "k_pool_spin" not in r["Name"])
    out["unlisted_kernels"] = {"us_per_slice": rest / nsl / 1e3}
    return out, f"profiles/{name} (rocprofv3 --kernel-trace of `bench.py --config5{' --si' if si else ''}{' --laser-solver multigrid' if solver == 'multigrid' else ''}`, {nsl} slices incl. the warm-up box; committed summary, not collected in this run)"


PHASE_OF_KERNEL = (("k_deposit_tiled", "deposit_current"), ("k_explicit_tiled", "explicit_deposit"), ("k_advance", "advance_plasma"),
                   ("k_dst_", "poisson"), ("k_tridiag", "poisson"), ("k_transpose", "poisson"), ("rocfft", "poisson"), ("k_dense", "poisson"),
                   ("k_smooth", "mg_solve1"), ("k_lower", "mg_solve1"), ("k_copy2", "mg_solve1"), ("k_post_norms", "mg_solve1"),
                   ("k_permute", "sort"), ("k_cell_keys", "sort"), ("k_rank_keys", "sort"), ("k_tile_", "sort"), ("k_run_starts", "sort"),
                   ("rocprim", "sort"))


def pmc_phase_bytes():
    """HBM bytes per slice of every PHASE of the slice (the engine's phase timers: deposit_current, poisson, explicit_deposit,
    mg_solve1, advance_plasma, sort, other) from the committed --pmc summary: the kernels of a phase by name, launches x
    (FETCH_SIZE x2 + WRITE_SIZE) / slices of that run, as pmc_slice_bytes.  None without a summary."""
    import csv
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    tot, nsl = {}, 0
    with open(path) as f:
        for r in csv.DictReader(f):
            try:
                b = (2.0 * float(r["FETCH_SIZE_raw_per_launch"]) + float(r["WRITE_SIZE_raw_per_launch"])) * 1024.0
            except ValueError:
                continue
            if b != b:
                continue
            ph = next((p for key, p in PHASE_OF_KERNEL if key in r["kernel"]), "other")
            tot[ph] = tot.get(ph, 0.0) + b * int(r["launches"])
            if r["kernel"].startswith("void hps::k_deposit_tiled<2, 16, 51"):
                nsl = int(r["launches"])
    return {k: v / nsl for k, v in tot.items()} if nsl else None


def cpu_baseline(n, ppc, nslices, threads):
    """Time the CPU oracle (restatement of the reference's CPU path) on the head `nslices` slices of the same deck:
    one thread (the reference's serial build) and `threads` OpenMP threads (4-colour tiles in the scatter kernels, as
    the reference's OpenMP build does: DepositionUtil.H:232-253)."""
    from hipace_amd import decks
    from oracle import oracle as O
    deck = decks.synthetic(n, 1024, ppc)
    host = os.cpu_count() or 1
    legs = {}
    for nt in sorted({1, max(1, threads)}):
        O.set_threads(nt)
        eng = O.Engine(deck)
        eng.begin_step()
        t0 = time.perf_counter()
        for k in range(nslices):
            eng.solve_slice(deck["nz"] - 1 - k)
        dt = time.perf_counter() - t0
        legs[nt] = (nslices / dt, dt)
        del eng
    O.set_threads(1)
    best = max(legs)
    out = dict(value=legs[best][0], unit="slices/s", cores=best, host_cores=host, kind="port",
               serial_value=legs[1][0],
               sample=f"oracle (C++ restatement of the reference's CPU path, g++ -O2 -fopenmp), head {nslices} slices of the same "
                      f"{n}x{n}x1024 {ppc * ppc}ppc deck: 1 thread {legs[1][1]:.1f} s"
                      + (f", {best} threads {legs[best][1]:.1f} s" if best > 1 else ""))
    # the head slices are the cheap ones (no sheath yet).  For scale, the same oracle over the WHOLE box of this deck, as timed
    # when the full-size parity fixture was written (another host: the build container): tests/golden/fullsize_config4.json
    try:
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_config4.json")))
        if n == 1024 and ppc == 2:
            out["whole_box_elsewhere"] = {"value": 1024.0 / fx["oracle_seconds"], "unit": "slices/s", "seconds": fx["oracle_seconds"],
                                          "what": fx["generated_by"] + ", all 1024 slices of the headline deck, on the build container's cores (not this host)"}
    except (OSError, KeyError, ValueError):
        pass
    return out


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, stdout=_JSON_OUT)       # the ranks inherit the real stdout (rank 0 prints the line)


_JSON_OUT = None


def claim_stdout():
    """Keep file descriptor 1 for the ONE JSON line: libraries under us write there too (RCCL prints a version banner
    from C stdio when a communicator is created).  Everything else that goes to stdout ends up on stderr."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    print(json.dumps(obj), file=_JSON_OUT or sys.stdout, flush=True)


def reference_benchmark(nxy):
    """examples/benchmarks/inputs_transverse_benchmark on one GPU (tests/transverse_benchmark.1Rank.sh runs it at nxy = 1023 and
    profiles/r05_reference_decks_host_beams.txt holds its checksums against the reference's file): slices/s of one whole box."""
    import torch
    from hipace_amd import api, decks
    deck = decks.transverse_benchmark(nxy, 1000)
    soa = decks.fixed_weight_pdf_beam(deck, seed=2024, **decks.TRANSVERSE_BENCHMARK_BEAM(nxy))
    eng = api.SliceEngine(deck, tile_size=16)
    eng.set_beam_particles(soa)
    nz = deck["nz"]
    times = []
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.begin_step()
        for isl in range(nz - 1, -1, -1):
            eng.solve_slice(isl)
        eng.sync()
        times.append(time.perf_counter() - t0)
    st = eng.stats()
    emit({
        "metric": f"transverse slices/s of the reference's transverse benchmark deck at {nxy}^2 x 1 ppc (explicit solver)", "value": nz / times[1],
        "unit": "slices/s", "n_gpus": 1, "steps": nz, "warmup": nz, "ms_per_step": 1e3 * times[1] / nz, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"examples/benchmarks/inputs_transverse_benchmark, my_constants.nxy = {nxy}: {nxy} x {nxy} x 1000 cells, 1 plasma "
                               f"electron per cell, fixed_weight_pdf beam of {soa.shape[1]} particles (drawn on the host by numpy), "
                               "absorbing particle boundary, hipace.dt = 0; NOT the BASELINE.json configuration (run bench.py without this flag)"},
        "vcycles_per_slice": st["vcycles"] / max(st["slices"], 1), "first_box_s": times[0], "timed_box_s": times[1]})
    return 0


def measure_handoff(transport, dev, rank, world, dist, torch, nbytes=1 << 20, reps=64):
    """What ONE hand-off costs on this edge, by itself: every rank posts `reps` receives of a 1 MiB block (a beam slice's block
    of the headline deck is 0.9 MB) and sends as many to the next rank, all ranks at once; seconds per message and the rate per
    edge -- between devices this is the xGMI link (ipc: a peer copy, rccl: ncclSend / ncclRecv), on one device a local copy.
    None on any failure (the leg's headline number does not depend on it)."""
    try:
        n = nbytes // 8
        rx = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(2)]
        tx = torch.full((n,), float(rank + 1), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        out = {}
        for label, count in (("warm", 8), ("timed", reps)):
            dist.barrier()
            t0 = time.perf_counter()
            evs = [transport.recv(rx[r & 1], None, slot=r % 64) for r in range(count)]
            for r in range(count):
                transport.send(tx, None, slot=r % 64)
            t1 = time.perf_counter()
            while not transport.ready(evs[-1]):
                if time.perf_counter() - t1 > 60.0:
                    raise RuntimeError("hand-off messages did not arrive within 60 s")
            transport.finish()
            out[label] = (time.perf_counter() - t0) / count
        transport._handoff_buffers = (rx, tx)      # (receive buffers stay allocated until the ring is destroyed: include/hpslice.h)
        ok = bool((rx[(reps - 1) & 1] == float((rank - 1) % world + 1)).all())
        worst = torch.tensor([out["timed"]], dtype=torch.float64)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        return {"bytes_per_message": nbytes, "messages": reps, "seconds_per_message": worst.item(), "GB_per_s_per_edge": nbytes / worst.item() / 1e9,
                "contents_checked": ok, "what": "every rank receives from its predecessor and sends to its successor at once, back to back (max over ranks)"}
    except Exception as exc:      # noqa: BLE001
        print(f"bench.py: rank {rank}: hand-off rate not measured ({type(exc).__name__}: {exc})", file=sys.stderr, flush=True)
        return None


def _pci_bus_id(dev):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    rc = hip.hipDeviceGetPCIBusId(buf, 64, int(dev))
    return buf.value.decode() if rc == 0 else f"hipDeviceGetPCIBusId={rc}"


def dry_leg_worker(args, rank, world):
    """--dry-legs: what a leg's worker does to the control flow, without a GPU -- join the leg's process group, agree on a
    number, rank 0 prints a line.  BENCH_DRY_FAIL=<kind>: rank 1 of that leg raises; BENCH_DRY_HANG=<kind>: it never comes back."""
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", init_method="file://" + os.environ["HPS_BENCH_LEG_RDZV"], rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=max(20.0, args.leg_timeout)))
    if rank == 1 and os.environ.get("BENCH_DRY_FAIL") == args.leg:
        raise RuntimeError(f"dry leg {args.leg}: rank 1 fails on purpose")
    if rank == 1 and os.environ.get("BENCH_DRY_HANG") == args.leg:
        time.sleep(3600)
    t = torch.ones(1)
    dist.all_reduce(t)
    if rank == 0:
        v = {"ipc": 2000.0, "rccl": 1900.0}[args.leg] * world
        emit({"metric": "dry run", "value": v, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": 1e3 * world / v, "ring_edge": args.leg, "ranks_seen": int(t.item()),
              "rccl_ranks_seen": int(t.item()) if args.leg == "rccl" else None, "value_steps_in_flight": 1.2 * v})
    dist.barrier()
    dist.destroy_process_group()
    return 0


def supervise_legs(args, rank, world):
    """N > 1: the process the launcher started for this rank does not touch the GPU.  It runs the measurement once per KIND OF
    EDGE between the ranks -- `ipc` (peer copies through hipIpc handles, ordered by a shared-memory mailbox) and `rccl`
    (ncclSend / ncclRecv, what the north star names) -- each as a worker process of its own (`bench.py ... --leg <kind>`), the
    workers of a leg meeting through a file store of their own.  A leg that fails, hangs or crashes is killed at --leg-timeout
    and recorded with its error text; the other leg's number is not lost with it, and no state of a failed leg (a stuck
    stream, a half-connected communicator) is carried into the next.  Rank 0 prints ONE line: the better leg's line, plus
    `ring_edges` = what every leg measured or why it could not, `rccl_ranks_seen` = the RCCL leg's count whichever leg won."""
    import signal
    import tempfile
    try:      # (and the supervisor goes when the launcher does, taking its worker with it: see die_with_parent)
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM, 0, 0, 0)
    except Exception:      # noqa: BLE001
        pass
    kinds = [args.edge] if args.edge else (["ipc"] if args.same_device else ["ipc", "rccl"])
    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a == "--edge":
            skip = True
            continue
        if a.startswith("--edge="):
            continue
        argv.append(a)
    key = "%s_%s_%d" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MASTER_PORT", "0"), os.getppid())
    key = "".join(c if c.isalnum() or c in "_-" else "_" for c in key)
    rdzv_dir = os.environ.get("HPS_BENCH_RDZV_DIR", tempfile.gettempdir())
    legs = {}
    for i, kind in enumerate(kinds):
        rdzv = os.path.join(rdzv_dir, f"hps_bench_rdzv_{key}_{i}_{kind}")
        env = dict(os.environ, HPS_RING_EDGE=kind, HPS_BENCH_LEG_RDZV=rdzv)
        env.setdefault("HPS_RING_TIMEOUT_S", "120")
        env.setdefault("HPS_RING_CONNECT_TIMEOUT_S", "120")
        errf = tempfile.TemporaryFile(mode="w+")
        t0 = time.perf_counter()
        def die_with_parent():
            # a worker must not outlive its supervisor (a driver that kills the launcher at ITS limit would otherwise leave the
            # workers, in sessions of their own, on the GPUs until their watchdogs fire): SIGKILL when the parent goes
            try:
                import ctypes
                ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL, 0, 0, 0)      # PR_SET_PDEATHSIG
            except Exception:      # noqa: BLE001
                pass

        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ["--leg", kind, "--watchdog", str(max(30.0, args.leg_timeout - 15.0))],
                             env=env, stdout=subprocess.PIPE, stderr=errf, text=True, start_new_session=True, preexec_fn=die_with_parent)
        timed_out = False
        try:
            out, _ = p.communicate(timeout=args.leg_timeout)
        except subprocess.TimeoutExpired:
            timed_out = True
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
            out, _ = p.communicate()
        errf.seek(0)
        err = errf.read()
        errf.close()
        if err:
            sys.stderr.write(f"---- bench.py rank {rank}, {kind} leg, stderr ----\n{err[-6000:]}\n")
            sys.stderr.flush()
        line = None
        for ln in (out or "").splitlines():
            if ln.startswith("{"):
                try:
                    line = json.loads(ln)
                except ValueError:
                    pass
        rec = {"seconds": time.perf_counter() - t0, "returncode": p.returncode}
        if timed_out:
            rec["error"] = f"the leg's worker of rank {rank} was still running after {args.leg_timeout:.0f} s and was killed"
        elif p.returncode != 0:
            last = [ln for ln in err.strip().splitlines() if ln.strip()][-3:]
            rec["error"] = f"the leg's worker of rank {rank} exited with code {p.returncode}: " + " | ".join(last)[-600:]
        legs[kind] = (rec, line)
        if timed_out or p.returncode != 0:
            time.sleep(3.0)          # (a killed worker's queues are torn down by the driver before the next leg starts)
    if rank != 0:
        # (a rank whose worker failed while rank 0's did not cannot say so through a group -- there is none between the
        #  supervisors by design; its stderr above holds the text, and rank 0's worker of that leg fails or times out with it)
        return 0
    for i, kind in enumerate(kinds):
        try:
            os.unlink(os.path.join(rdzv_dir, f"hps_bench_rdzv_{key}_{i}_{kind}"))
        except OSError:
            pass
    edges = {}
    for kind in ("ipc", "rccl"):
        if kind not in legs:
            edges[kind] = {"value": None, "skipped": ("RCCL refuses two ranks on one device (--same-device)" if (args.same_device and kind == "rccl")
                                                      else f"--edge {args.edge} asked for the other kind only")}
            continue
        rec, line = legs[kind]
        e = {"value": line.get("value") if line else None,
             "value_steps_in_flight": line.get("value_steps_in_flight") if line else None,
             "ms_per_step": line.get("ms_per_step") if line else None,
             "ranks_seen": line.get("ranks_seen") if line else None,
             "leg_seconds": rec["seconds"]}
        if kind == "rccl":
            e["rccl_ranks_seen"] = line.get("rccl_ranks_seen") if line else None
        if line and line.get("rank_devices"):
            e["rank_devices"] = line["rank_devices"]
        if line and line.get("handoff"):
            e["handoff"] = line["handoff"]
        if line and isinstance(line.get("in_flight"), dict) and line["in_flight"].get("error"):
            e["in_flight_error"] = line["in_flight"]["error"]
        if "error" in rec:
            e["error"] = rec["error"]
        elif line is None:
            e["error"] = "the leg's rank-0 worker ended without a line"
        edges[kind] = e
    good = [k for k in kinds if legs[k][1] is not None and legs[k][1].get("value")]
    if not good:
        emit({"metric": "transverse slices/s", "value": None, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "error": "no ring-edge leg produced a number", "ring_edges": edges})
        return 1
    best = max(good, key=lambda k: legs[k][1]["value"])
    out = dict(legs[best][1])
    out["ring_edge"] = best
    out["ring_edges"] = edges
    out["value_is_of_edge"] = best
    rc = edges.get("rccl", {})
    out["rccl_ranks_seen"] = rc.get("rccl_ranks_seen")
    if out["rccl_ranks_seen"] != world:
        out["rccl_ranks_seen_note"] = rc.get("error") or rc.get("skipped") or "the RCCL leg ran but did not count every rank on its edges"
    emit(out)
    return 0


def main():
    claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed slices (per GPU); 0 = one whole box (1024 slices; --config5: 2048, --config2: 512)")
    ap.add_argument("--warmup", type=int, default=64, help="untimed warm-up slices")
    ap.add_argument("--n", type=int, default=1024, help="transverse cells per side")
    ap.add_argument("--ppc", type=int, default=2, help="plasma particles per cell per direction")
    ap.add_argument("--tile", type=int, default=16, help="particle tile size (0, 16, 32)")
    ap.add_argument("--sort-period", type=int, default=128, help="max slices between particle re-sorts (adaptive below)")
    ap.add_argument("--cpu-slices", type=int, default=4, help="slices of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help=f"threads of the CPU baseline's OpenMP leg (0 = min(host cores, {CPU_THREADS_DEFAULT}))")
    ap.add_argument("--start-slice", type=int, default=-1,
                    help="--steps < nz: slice (counted from the head) where the warm-up + timed window starts; -1 = "
                         f"{START_SLICE_DEFAULT} scaled to the box (a window there costs what the whole box costs on average)")
    ap.add_argument("--profile-stride", type=int, default=0,
                    help="HIP-event phase timers (and with them the roofline's kernel duration) on every n-th slice of the "
                         "timed region: the 11 event records of a timed slice cost 4.5 %% of it (whole boxes: 1474 / 1482 / 1489 slices/s with "
                         "every 7th / 32nd / 128th slice timed).  0 = 16 for whole boxes, 2 "
                         "when fewer than 64 slices are timed (4 event records per timed slice there: 1464 / 1476 / 1485 slices/s with "
                         "every / every 2nd / every 4th of 20 slices timed)")
    ap.add_argument("--phase-window", type=int, default=32,
                    help="--steps < nz, N = 1: slices of an extra window behind the timed region that is run with the full phase timers "
                         "(11 event records per slice) to fill phase_ms_per_slice / roofline.per_kernel; 0 = none")
    ap.add_argument("--inflight", type=int, default=3,
                    help="second measurement: L time steps in flight on the GPU (hipace_amd/pipeline.py::run_lanes: L engines on L "
                         "streams driven by one host thread, step s+1 trails step s by the per-slice beam hand-off) -> "
                         "value_steps_in_flight; `value` is always the one-engine number.  1 = skip.  N > 1: L stages per rank on "
                         "the ring (world x L stages)")
    ap.add_argument("--inflight-ring", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--one-stage-per-rank", action="store_true",
                    help="N > 1: skip the second measurement (--inflight stages per rank on the ring, world x L stages)")
    ap.add_argument("--laser-solver", choices=["fft", "multigrid"], default="fft",
                    help="--config5: lasers.solver_type (multigrid = hpmg system type 2, the reference's default)")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE config 5: laser_blowout_wake 1024x1024x2048, 4 ppc, a Gaussian laser pulse drives the wake "
                         "and is advanced by the envelope solver on every slice; the time levels of the envelope stay in HBM "
                         "(not the judged bench line)")
    ap.add_argument("--handoff-batch", type=int, default=1,
                    help="slices per beam hand-off group on the ring (hipace_amd.pipeline.run_pipeline); 1 = one hand-off per slice")
    ap.add_argument("--ring-self", action="store_true",
                    help="one GPU, but every slice's hand-off goes through the RCCL ring (hipace_amd.pipeline.RcclSelfRing): the "
                         "multi-rank code path with the rank as its own neighbour -- what the ring costs per slice.  Needs "
                         "--steps >= 2 boxes")
    ap.add_argument("--fuse", action="store_true",
                    help="fused schedule: the push of slice k also deposits the currents of slice k-1 (hps_engine_set_fusion)")
    ap.add_argument("--no-ionization", action="store_true", help="--config5 without the ionisable species")
    ap.add_argument("--si", action="store_true",
                    help="--config5 in SI units (hipace.normalized_units = 0: the deck of tests/laser_blowout_wake_explicit.SI.1Rank.sh, as "
                         "BASELINE.json names configs[4]) instead of its normalised twin")
    ap.add_argument("--config2", action="store_true",
                    help="BASELINE config 2 instead of the headline workload: linear_wake 256x256x512, 4 ppc, "
                         "predictor-corrector Bx/By solver (not the judged bench line)")
    ap.add_argument("--watchdog", type=float, default=1500.0,
                    help="N > 1: seconds after which a rank that is still running gives up with exit code 3 (a ring that "
                         "cannot connect must not hold the node)")
    ap.add_argument("--edge", choices=["ipc", "rccl"], default=None,
                    help="N > 1 (and --ring-self): kind of edge between processes -- ipc (default): peer copies into the next rank's "
                         "buffers, ordered through a shared-memory mailbox (hps_ring_*, HPS_RING_EDGE); rccl: ncclSend / ncclRecv")
    ap.add_argument("--same-device", action="store_true",
                    help="N > 1: every rank uses device 0 -- N processes, each with its own engine(s), hand over through the ipc "
                         "edge on ONE GPU (what a one-GPU box can run of the multi-process path; RCCL refuses two ranks per device)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="only check the launch path: every rank joins the process group (gloo, no GPU needed) and rank 0 "
                         "prints how many ranks there are")
    ap.add_argument("--leg", choices=["ipc", "rccl"], default=None,
                    help="(internal, N > 1) this process is the worker of one ring-edge leg: bench.py under a launcher starts one "
                         "worker per rank and edge kind, see supervise_legs")
    ap.add_argument("--leg-timeout", type=float, default=600.0,
                    help="N > 1: seconds one edge kind's leg (all its ranks) may take before its workers are killed and the leg is "
                         "recorded as failed; the other kind's number is not lost with it")
    ap.add_argument("--dry-legs", action="store_true",
                    help="N > 1, no GPU: the workers only join their process group and print a made-up line (control flow of the "
                         "two legs; BENCH_DRY_FAIL=<kind> makes that leg's rank 1 fail, BENCH_DRY_HANG=<kind> makes it hang)")
    ap.add_argument("--reference-benchmark", type=int, nargs="?", const=1023, default=0, metavar="NXY",
                    help="instead of the BASELINE deck: the reference's own transverse scaling benchmark "
                         "(examples/benchmarks/inputs_transverse_benchmark: NXY^2 x 1000 cells, 1 ppc, a fixed_weight_pdf beam of "
                         "10 NXY^2 particles drawn on the host; default NXY = 1023, the size its CI runs) -- one untimed box, one timed")
    args = ap.parse_args()
    if args.reference_benchmark:
        return reference_benchmark(args.reference_benchmark)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(respawn_under_torchrun(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        assert os.environ.get("HPS_RING_EDGE", "ipc") != "rccl", "--same-device needs the ipc edge (RCCL refuses two ranks on one device)"
        local = 0
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")

    if world > 1 and args.leg is None and not args.spawn_check:
        return supervise_legs(args, rank, world)

    import torch
    import torch.distributed as dist
    if args.spawn_check:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            emit({"spawn_check": True, "n_gpus": int(t.item()), "world_size": dist.get_world_size()})
        dist.destroy_process_group()
        return

    if args.dry_legs and args.leg is not None:
        return dry_leg_worker(args, rank, world)

    from hipace_amd import api, decks
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    ctl = None
    if world > 1:
        import threading

        progress = {"phase": "start", "slice": -1, "transport": None}

        def give_up():
            # fail loudly: where this rank was and what its ring has carried so far, then exit (the driver sees code 3)
            st = None
            try:
                st = progress["transport"].stats() if progress["transport"] is not None else None
            except Exception as exc:      # noqa: BLE001
                st = f"unavailable ({exc})"
            print(f"bench.py: rank {rank} of {world} still running after {args.watchdog:.0f} s -- giving up; phase {progress['phase']}, "
                  f"last slice enqueued {progress['slice']}, ring {st}", file=sys.stderr, flush=True)
            os._exit(3)

        dog = threading.Timer(args.watchdog, give_up)
        dog.daemon = True
        dog.start()
        # torch.distributed is the bootstrap and the clock's barrier only (the ids of the ring's edges, max-over-ranks of
        # the times): a host-only gloo group -- one node, loopback.  The data path is the C-ABI ring.  (Fallback: torch's RCCL
        # group, whose barrier is a kernel plus a wait that may synchronise the whole device -- not usable with --same-device.)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        try:
            import datetime
            rdzv = os.environ.get("HPS_BENCH_LEG_RDZV")
            if rdzv:      # a leg's workers meet through a file of their own (the launcher's store belongs to the supervisors)
                dist.init_process_group("gloo", init_method="file://" + rdzv, rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=max(60.0, args.leg_timeout)))
            else:
                dist.init_process_group("gloo")
            ctl = dist.group.WORLD
        except Exception as exc:      # noqa: BLE001
            print(f"bench.py: no gloo process group ({exc}); using torch's RCCL group", file=sys.stderr)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            ctl = None
    red_dev = "cpu" if ctl is not None else "cuda"

    def meet_mid_run():
        """all ranks have arrived -- without a device-wide synchronise (see on_slice below)"""
        if ctl is not None:
            dist.barrier(group=ctl)
        else:
            t = torch.zeros(1, device="cuda")
            dist.all_reduce(t)
            torch.cuda.current_stream().synchronize()

    def reduce_over_ranks(x, op):
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=op)
        return t.item()

    nz = 1024
    deck = decks.synthetic(args.n, nz, args.ppc)
    if args.config2:
        nz, args.n, args.ppc = 512, 256, 2
        deck = decks.predictor_corrector(decks.linear_wake(), 4.0e-2, 30, 0.05)
        deck.update(nx=256, ny=256, nz=nz, plasma_ppc=(2, 2))
        args.cpu_slices = 0
    if args.config5:
        nz = 2048
        # neutral nitrogen, a fifth of the electron density, one macro-atom per cell: ionised by the wake (ADK); --si: the deck in
        # SI units as BASELINE names it, else its normalised twin with kp_inv = 10 um (hipace.background_density_SI)
        deck = decks.config5(args.n, nz, 2 if args.laser_solver == "multigrid" else 1, si=args.si, ionize=not args.no_ionization)
        args.cpu_slices = 0
        args.inflight = 1
    if args.steps <= 0:
        args.steps = nz                         # one whole box of the deck that is timed (config 5: all 2048 slices, the pulse included)
    if world > 1 and args.one_stage_per_rank:
        args.inflight = 1
    if args.config2 and os.environ.get("HPS_PC_SPECULATE", "1") == "0":
        args.inflight = 1                       # (host-controlled predictor-corrector loop: the host is held once per iteration, the slice
                                                #  has no enqueue-only first half for a one-thread driver of several engines to interleave)
    eng = api.SliceEngine(deck, device=local, tile_size=args.tile, sort_period=args.sort_period)
    lanes = 1                                   # the headline measurement: one engine
    engines = [eng]
    if args.fuse:
        for e in engines:
            e.set_fusion(True)
    short = args.steps < nz and lanes == 1 and not args.ring_self
    stride = args.profile_stride if args.profile_stride > 0 else (2 if args.steps < 64 else 16)
    dev = torch.device("cuda", local)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def profiling(on):
        # short runs time every 2nd slice: only the 4 event records the roofline needs (11 would cost 4.5 % of a slice)
        for e in engines:
            e.set_profiling(on, stride=stride, light=short)

    transport = None
    handoff = None
    if world > 1:
        from hipace_amd.pipeline import RingTransport
        progress["phase"] = "ring init"
        # the edges are connected before the clock starts.  A leg's edge kind is fixed (no fall-back inside a leg: the other
        # kind has a leg of its own) and proven first: one 4 KB message to the next rank and one from the previous rank, contents
        # checked -- between two DEVICES this is the first use of the peer mapping (ipc) / the communicators (RCCL)
        transport = RingTransport(rank, world, local, edge=args.leg)
        progress["transport"] = transport
        if args.leg is not None:
            progress["phase"] = "ring probe"
            if not transport._probe(local, seconds=60.0):
                raise RuntimeError(f"rank {rank}: the {transport.kind} edge's probe message failed: {transport._probe_error}")
            progress["phase"] = "hand-off rate"
            handoff = measure_handoff(transport, dev, rank, world, dist, torch)
        progress["phase"] = "headline run"
    elif args.ring_self:
        from hipace_amd.pipeline import RcclSelfRing, ring_edge
        transport = RcclSelfRing(local, edge=ring_edge())

    clock = {}
    phases_timed = phase_window = None
    stats0 = {}
    timed_first = 0
    if short:
        # one step per rank; untimed down to the representative region, then warm-up, then K timed slices
        start = args.start_slice if args.start_slice >= 0 else (START_SLICE_DEFAULT * nz) // 1024
        start = max(0, min(start, nz - args.steps - args.warmup))
        lead = start + args.warmup                              # untimed slices of rank 0
        # rank r runs `lag` slices behind rank r-1: it needs the beam of its next slice, which arrives with that
        # slice's hand-off group (run_pipeline's handoff_batch), and one slice of slack on top
        lag = 2 if args.handoff_batch <= 1 else args.handoff_batch + 2
        lead = max(lead, lag * (world - 1))                     # every rank needs a non-negative untimed part
        assert lead + args.steps <= nz, "window does not fit the box"
        counts = [lead - lag * r + args.steps for r in range(world)]
        first = lead - lag * rank
        timed_first = first

        def on_slice(m, q):
            if world > 1:
                progress["slice"] = q
            if q == first:
                if transport is not None:
                    # mid-run, pipeline filled: this rank's receives of the slices to come are posted (a whole step
                    # ahead, so that the previous rank's sends never wait for this one) and their RCCL kernels sit on
                    # the ring's receive stream until the data comes -- a device-wide synchronise would wait for
                    # messages the previous rank only sends after this barrier.  Synchronise what the clock is about:
                    # the engine's stream and the sends.
                    eng.sync()
                    transport.sync_sends()
                    meet_mid_run()
                else:
                    barrier()
                stats0.update(eng.stats())
                profiling(True)
                # (the host reads the multigrid's norms once per slice: a collection pass of the interpreter inside a 12 ms window
                #  is a per-cent-level hiccup -- one run in ~40 showed 1283 instead of 1700; no collection while the clock runs)
                gc.freeze()
                gc.disable()
                clock["t0"] = time.perf_counter()
            elif q == first + args.steps:
                eng.sync()
                if transport is not None:
                    transport.sync_sends()
                clock["t1"] = time.perf_counter()
                gc.enable()
                gc.unfreeze()

        if world == 1:
            eng.begin_step()
            for q in range(counts[0]):
                on_slice(0, q)
                eng.solve_slice(nz - 1 - q)
            on_slice(0, counts[0])
        else:
            from hipace_amd.pipeline import run_pipeline
            run_pipeline(eng, rank, world, world, dev, slices_per_step=counts, transport=transport, on_slice=on_slice,
                         handoff_batch=args.handoff_batch)
        barrier()
        dt = clock["t1"] - clock["t0"]
        if world == 1 and args.phase_window > 0 and counts[0] + args.phase_window <= nz:
            # the driver's short run times the slices with the light timers (the deposition only: 4 event records per timed
            # slice); the split of a slice over its phases comes from one more window AFTER the timed region, every slice of
            # it with all 11 event records -- not part of `value`
            phases_timed = eng.phase_times()
            eng.set_profiling(True, stride=1, light=False)
            for q in range(counts[0], counts[0] + args.phase_window):
                eng.solve_slice(nz - 1 - q)
            eng.sync()
            phase_window = dict(zip(("ms", "slices"), eng.phase_times()), first=counts[0], last=counts[0] + args.phase_window - 1)
            eng.set_profiling(True, stride=stride, light=True)       # (cleared: phase_times below reads what is kept here)
    else:
        def run_slices(count):
            done = 0
            while done < count:
                eng.begin_step()
                m = min(nz, count - done)
                for k in range(m):
                    eng.solve_slice(nz - 1 - k)
                done += m

        if args.config5:
            # an evolving envelope (hipace.dt != 0) must not be warmed up on the engine that is timed: a second
            # begin_step would rotate the time levels of a step that solved only the warm-up slices, and the timed box
            # would hold no pulse (that is what the round-1 config-5 lines measured).  Warm the kernels on a short box.
            warm = api.SliceEngine(dict(deck, nz=max(args.warmup, 8)), device=local, tile_size=args.tile, sort_period=args.sort_period)
            warm.begin_step()
            for k in range(max(args.warmup, 8)):
                warm.solve_slice(max(args.warmup, 8) - 1 - k)
            warm.sync()
            del warm
        else:
            run_slices(args.warmup)
        barrier()
        stats0.update(eng.stats())
        profiling(True)
        gc.freeze()
        gc.disable()
        t0 = time.perf_counter()
        if world == 1 and args.ring_self:
            from hipace_amd.pipeline import run_pipeline
            stamps = []
            tl = (lambda m, q: stamps.append((m, q, time.perf_counter())) if q % 64 == 0 else None) if os.environ.get("BENCH_TIMELINE") else None
            args.steps = run_pipeline(eng, 0, 1, max(2, args.steps // nz), dev, transport=transport, handoff_batch=args.handoff_batch,
                                      on_slice=tl)
            for a, b in zip(stamps, stamps[1:]):
                print(f"timeline step {a[0]} slice {a[1]:4d}: {1e3 * (b[2] - a[2]) / 64:.4f} ms/slice", file=sys.stderr)
        elif world == 1:
            run_slices(args.steps)
        else:
            # ring pipeline over time steps: every rank sweeps whole boxes; the beam slices travel rank -> rank+1
            from hipace_amd.pipeline import run_pipeline
            steps_per_rank = max(1, args.steps // nz)
            args.steps = run_pipeline(eng, rank, world, world * steps_per_rank, dev, transport=transport, handoff_batch=args.handoff_batch)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        gc.unfreeze()
    phases, nprof = phases_timed if phases_timed is not None else eng.phase_times()
    eng.set_profiling(False)
    for e in engines[1:]:            # phase times: mean over the lanes (intervals overlap in wall time)
        ph, n = e.phase_times()
        e.set_profiling(False)
        for k in phases:
            phases[k] += ph[k]
        nprof += n
    ring_stats = transport.stats() if transport is not None else None
    rccl_ranks_seen = None
    if transport is not None:
        # evidence that N processes were on the ring: what this rank's two ring edges report about themselves (RCCL: ncclCommCount
        # = 2 each on a ring of 2+ ranks; ipc: both ends of the edge's mailbox attached) and that messages went both ways;
        # summed over the ranks below
        inf = transport.info()
        ring_stats = dict(ring_stats, **inf)
        # (an open ring -- one step per rank, the short run -- has a head rank that only sends and a tail rank that only receives)
        on_ring = (inf["comm_in_ranks"] == 2 and inf["comm_out_ranks"] == 2 and ring_stats["sent"] + ring_stats["received"] > 0) if world > 1 \
            else (inf["comm_out_ranks"] == 1 and ring_stats["sent"] > 0)
        rccl_ranks_seen = int(on_ring)
        if world > 1:
            rccl_ranks_seen = int(round(reduce_over_ranks(rccl_ranks_seen, dist.ReduceOp.SUM)))
    rank_devices = None
    if world > 1:
        dt = reduce_over_ranks(dt, dist.ReduceOp.MAX)
        # which device every rank ran on (PCI bus id as the runtime reports it): the evidence that the edges crossed devices
        try:
            mine = f"{local}:{torch.cuda.get_device_properties(local).name}:{_pci_bus_id(local)}"
        except Exception as exc:      # noqa: BLE001
            mine = f"{local}:?({exc})"
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
    st_headline = eng.stats()
    snap = dict(laser_vc=eng.laser_vcycles() if args.config5 else None, pc=eng.pc_stats()[0] if args.config2 else None,
                sorts=eng.sorts() if args.tile else 0, fallbacks=eng.fallbacks() if args.tile else 0,
                ion=eng.ion_stats() if (args.config5 and not args.no_ionization) else None)

    # ---- second measurement: L time steps in flight per GPU (pipeline.run_lanes) --------------------------------------
    if world > 1:
        progress["phase"] = "steps-in-flight run"
    inflight = None
    L = max(1, args.inflight)

    def measure_in_flight():
        from hipace_amd.pipeline import run_lanes
        lane_engines = [eng] + [api.SliceEngine(deck, device=local, tile_size=args.tile, sort_period=args.sort_period) for _ in range(L - 1)]
        if args.fuse:
            for e in lane_engines[1:]:
                e.set_fusion(True)
        W = world * L
        lagL = 2                                                 # a stage trails the one ahead by the hand-off of two slices
        if args.steps < nz:
            # steady-state window: every stage runs (untimed) to its place in the box -- stage v `lagL*v` slices behind stage
            # 0 --, then all of them time `steps` slices at once.  The clock is the device's: an event on each engine's
            # stream at the first and behind the last timed slice (one host thread drives the stages and must not stop).
            startL = args.start_slice if args.start_slice >= 0 else (START_SLICE_DEFAULT * nz) // 1024
            leadL = max(min(startL, nz - args.steps - args.warmup) + args.warmup, lagL * (W - 1))
            assert leadL + args.steps <= nz, "window does not fit the box"
            countsL = [leadL - lagL * v + args.steps for v in range(W)]
            streams = [torch.cuda.ExternalStream(e.stream_handle(), device=dev) for e in lane_engines]
            ref_ev = torch.cuda.Event(enable_timing=True)
            ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(L)]
            ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(L)]
            barrier()
            ref_ev.record(streams[0])

            def on_slice_L(j, m, q):
                first = leadL - lagL * (rank * L + j)
                if q == first:
                    ev0[j].record(streams[j])
                elif q == first + args.steps:
                    ev1[j].record(streams[j])

            run_lanes(lane_engines, rank, world, W, dev, slices_per_step=countsL, transport=transport, on_slice=on_slice_L)
            barrier()
            t_start = min(ref_ev.elapsed_time(e) for e in ev0)
            t_end = max(ref_ev.elapsed_time(e) for e in ev1)
            dtL = 1e-3 * (t_end - t_start)
            nL = args.steps
            whole = False
        else:
            boxes = max(1, args.steps // nz)
            run_lanes(lane_engines, rank, world, W, dev, slices_per_step=max(2, min(args.warmup, nz)), transport=transport)   # warm every stage
            barrier()
            t0 = time.perf_counter()
            if os.environ.get("HPS_DRIVE_TRACE"):
                from hipace_amd import pipeline as _pl
                del _pl._TRACE[:]
            solvedL = run_lanes(lane_engines, rank, world, W * boxes, dev, transport=transport)
            if os.environ.get("HPS_DRIVE_TRACE"):
                _pl.dump_trace()
            barrier()
            dtL = time.perf_counter() - t0
            nL = solvedL // L
            whole = True
        if world > 1:
            dtL = reduce_over_ranks(dtL, dist.ReduceOp.MAX)
        inflight = dict(value=W * nL / dtL, stages_per_gpu=L, slices_per_stage=nL, seconds=dtL,
                        window="whole boxes, pipeline fill included" if whole else
                               f"{nL} slices per stage in steady state (device clock: events on the stages' streams)")
        if not (args.config5 or args.config2):
            # the GPU's time per slice with L stages sharing it, against the survey's fused lower bound and the counter traffic
            nvL = sum(e.stats()["vcycles"] for e in lane_engines) / max(sum(e.stats()["slices"] for e in lane_engines), 1)
            _, b_fusedL = slice_bytes(args.n, args.ppc * args.ppc, nvL)
            t_gpu = dtL * world / (W * nL)
            cb = pmc_slice_bytes() if (args.tile == 16 and args.n == 1024 and args.ppc == 2) else None
            inflight["roofline"] = {
                "seconds_per_slice_of_the_gpu": t_gpu, "vcycles": nvL,
                "algorithmic_bytes_fused_lower_bound": b_fusedL, "achieved_fused": b_fusedL / t_gpu / 1e9,
                "frac_fused_lower_bound": b_fusedL / t_gpu / 1e9 / HBM_PEAK_GBS,
                "counter_bytes_per_slice": cb, "achieved_counter": cb / t_gpu / 1e9 if cb else None,
                "frac_counter_bytes": cb / t_gpu / 1e9 / HBM_PEAK_GBS if cb else None,
                "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "note": "one stage's counter bytes per slice (profiles/" + PMC_SUMMARY + ") over the GPU's time per slice with all stages "
                        "in flight; the aggregate counter run of the L-stage window is profiles/r04g_inflight_pmc.csv"}
        del lane_engines[1:]
        return inflight

    if L > 1:
        try:
            inflight = measure_in_flight()
        except Exception as exc:      # noqa: BLE001
            # N > 1: the headline number must not be lost with the second measurement (a ring that times out raises here)
            if world == 1:
                raise
            import traceback
            traceback.print_exc()
            inflight = dict(value=None, stages_per_gpu=L, error=f"{type(exc).__name__}: {exc}")

    if rank == 0:
        total = args.steps * world
        st1 = st_headline
        nsl = max(st1["slices"] - stats0.get("slices", 0), 1)
        ab = algorithmic_bytes(args.n, args.ppc * args.ppc)
        per_kernel = {k: phases[k] / max(nprof, 1) for k in phases}
        dom = "deposit_current"
        # the interval between two HIP events around ONE kernel = the kernel + what an interval costs by itself; the
        # schedule has one interval without any kernel (two records back to back), measured on the same slices: subtract
        # it.  (rocprofv3's kernel duration is begin -> end of the kernel alone.)
        overhead = per_kernel.pop("empty_interval", 0.0)
        raw = per_kernel[dom]
        kernel_ms = max(raw - overhead, 0.0) if lanes == 1 else raw
        if short:       # light timers: only the deposition was timed; the rest of the split from the window behind the timed region
            per_kernel = {k: (v if k == dom else None) for k, v in per_kernel.items()}
            if phase_window is not None:
                pw = {k: v / max(phase_window["slices"], 1) for k, v in phase_window["ms"].items()}
                ov_w = pw.pop("empty_interval", 0.0)
                for k in per_kernel:
                    if k != dom:
                        per_kernel[k] = pw.get(k)
        achieved = ab[dom] / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        headline = args.tile == 16 and args.n == 1024 and args.ppc == 2 and not args.config5 and not args.config2
        traffic, traffic_source = pmc_traffic("void hps::k_deposit_tiled<2, 16, 51") if headline else (None, None)
        out = {
            "metric": "transverse slices/s at 256^2 x 4ppc (predictor-corrector solver)" if args.config2 else
                      f"transverse slices/s at {args.n}^2 x 4ppc with a laser envelope (explicit solver, {args.laser_solver} envelope solver)" if args.config5 else
                      f"transverse slices/s at {args.n}^2 x {args.ppc * args.ppc}ppc (explicit solver)",
            "value": total / dt, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "particle_pushes_per_s": total / dt * args.ppc * args.ppc * args.n * args.n,
            "config": {"workload": ("linear_wake.normalized 256x256x512, 4 ppc, order 2, predictor-corrector Bx/By solver "
                                    "(tolerance 4e-2, <= 30 iterations, mixing 0.05), dt=0 (BASELINE.json configs[1])") if args.config2 else
                                   (f"laser_blowout_wake_explicit{'.SI' if args.si else ''} {args.n}x{args.n}x{nz} in {'SI units (hipace.normalized_units = 0, kp_inv = 10 um: the deck BASELINE names)' if args.si else 'NORMALISED units (BASELINE names the .SI twin of the deck: same kernels, other constants; --si runs that one)'}, 4 ppc, order 2, Gaussian laser a0=4.5 + {args.laser_solver} envelope "
                                    "solver every slice" + (", no ionisation" if args.no_ionization else ", neutral N (0.2 n_e, 1 macro-atom per cell) "
                                    "field-ionised by the wake (ADK), released electrons join the plasma") + " (BASELINE.json configs[4])") if args.config5 else
                                   (f"blowout_wake synthetic {args.n}x{args.n}x{nz}, {args.ppc * args.ppc} ppc, "
                                    "order 2, explicit Bx/By solver, dt=0 (BASELINE.md section 3)"),
                       "parallelism": f"time-step pipeline x{world}; `value`: one time step per GPU at a time, `value_steps_in_flight`: "
                                      f"{max(1, args.inflight)} per GPU"},
            "timed_slices": ({"first": timed_first, "last": timed_first + args.steps - 1, "counted_from": "head of the box",
                              "pipeline_prefilled": world > 1,
                              # N > 1: the clock starts with the pipeline filled, so `value` is a steady-state rate, not
                              # nz*nsteps / wall(Evolve): the last rank starts fill_slices slices after the first, and a whole
                              # run of one step per rank would come out at value_including_fill
                              "fill_slices": lag * (world - 1) if world > 1 else 0,
                              "value_including_fill": (total / dt) * nz / (nz + lag * (world - 1)) if world > 1 else None} if short else
                             {"whole_boxes": max(1, args.steps // nz), "slices_per_box": nz, "pipeline_prefilled": False}),
            "steps_in_flight": inflight["stages_per_gpu"] if inflight else 1,
            "value_steps_in_flight": inflight["value"] if inflight else None,
            "in_flight": inflight,
            "fused_push_deposit": bool(args.fuse),
            "phase_ms_per_slice": per_kernel,
            "profiled_slices": nprof,
            "phase_window": ({"first": phase_window["first"], "last": phase_window["last"], "slices": phase_window["slices"],
                              "what": "behind the timed region, full phase timers on every slice; fills phase_ms_per_slice except deposit_current"}
                             if phase_window is not None else None),
            "vcycles_per_slice": (st1["vcycles"] - stats0.get("vcycles", 0)) / nsl,
            "laser_vcycles_per_slice": (snap["laser_vc"] / max(st1["slices"], 1)) if args.config5 else None,
            "pc_iterations_per_slice": snap["pc"] / max(st1["slices"], 1) if args.config2 else None,
            "particle_sorts": snap["sorts"],
            "halo_fallbacks": snap["fallbacks"],
            "ionization": (dict(zip(("electrons_released", "product_species_particles"), snap["ion"]))
                           if (args.config5 and not args.no_ionization) else None),
            "ring": ring_stats,
            "ring_edge": (transport.kind if transport is not None else None),
            "ranks_seen": rccl_ranks_seen,                  # ranks whose two ring edges are connected to a neighbour and carried messages both ways
            "rccl_ranks_seen": rccl_ranks_seen if (transport is not None and transport.kind == "rccl") else None,
            "ranks_on_one_device": bool(args.same_device) if world > 1 else None,
            "rank_devices": rank_devices,
            "handoff": handoff,
            "stages_per_rank_on_the_ring": (max(1, args.inflight) if (world > 1 or args.ring_self) else None),
            "roofline": {"bound": "hbm", "kernel": "k_deposit_tiled<2,%d>" % args.tile if args.tile else "k_deposit_current<2>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": kernel_ms,
                         # SURVEY 8(d)'s figure reads idcpu (56 B per particle); this build's kernel takes "weight != 0" for the valid
                         # bit (HPS_VALID_BY_W, default on) and reads 48: the same kernel time over the bytes it has to move itself
                         "bytes_this_build_has_to_move": (ab[dom] - 8 * args.ppc * args.ppc * args.n * args.n) if (headline and os.environ.get("HPS_VALID_BY_W", "1") != "0") else ab[dom],
                         "frac_of_bytes_this_build_has_to_move": ((ab[dom] - 8 * args.ppc * args.ppc * args.n * args.n) if (headline and os.environ.get("HPS_VALID_BY_W", "1") != "0") else ab[dom])
                                                                 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0,
                         "event_interval_ms": raw, "empty_event_interval_ms": overhead,
                         "duration_source": f"HIP events on the engine's stream around the kernel, {nprof} launches of the timed region, "
                                            "minus the interval between two back-to-back event records measured on the same slices"},
        }
        if not (args.config5 or args.config2) and all(per_kernel.get(k) is not None for k in ("poisson", "explicit_deposit", "mg_solve1", "advance_plasma")):
            # every phase of the slice against the HBM roofline, not only the deposition: time per slice (HIP events around the
            # phase, one empty event interval subtracted), SURVEY 8(d)'s algorithmic bytes, the committed counter bytes
            nv_ = (st1["vcycles"] - stats0.get("vcycles", 0)) / nsl
            C_ = args.n * args.n
            alg = dict(ab, mg_solve1=17 * 8 * C_ + (4.0 / 3.0) * 23 * 8 * C_ * nv_)
            cbp = pmc_phase_bytes() if headline else None
            ov = overhead if not short else (ov_w if phase_window is not None else 0.0)
            rows = {}
            for k, what in (("advance_plasma", "k_advance_tiled (gather + push)"), ("explicit_deposit", "k_explicit_tiled"),
                            ("deposit_current", "k_deposit_tiled"), ("poisson", "3 DST solves (k_dst_rows*), sources formed in the first pass"),
                            ("mg_solve1", "hpmg solve1: all k_smooth / k_lower launches of the Bx, By solve")):
                ms = max(per_kernel[k] - (ov if lanes == 1 else 0.0), 0.0)
                if k == dom:
                    ms = kernel_ms
                cbk = cbp.get(k) if cbp else None
                rows[k] = {"kernels": what, "us_per_slice": 1e3 * ms, "algorithmic_bytes": alg[k],
                           "achieved": alg[k] / (ms * 1e-3) / 1e9 if ms > 0 else None,
                           "frac": alg[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None,
                           "counter_bytes": cbk, "frac_counter_bytes": cbk / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if (cbk and ms > 0) else None}
                if k == "mg_solve1" and ms > 0:
                    # SURVEY 8(d)'s bytes are those of the REFERENCE's pass structure; the fused smoothers move about half of
                    # them.  `frac` is what the kernels move (counter bytes) over their time; the survey-bytes figure is kept
                    # beside it under its own name
                    rows[k]["frac_reference_passes"] = rows[k]["frac"]
                    if cbk:
                        rows[k]["achieved"] = cbk / (ms * 1e-3) / 1e9
                        rows[k]["frac"] = rows[k]["frac_counter_bytes"]
                        rows[k]["frac_is_of"] = "counter bytes (what the kernels move); frac_reference_passes: SURVEY 8(d)'s bytes of the reference's passes"
                if k == "poisson" and ms > 0:
                    # SURVEY 8(d) prices a solve at 4 transform passes (16 B per cell each); this build makes 3 passes -- DST along
                    # x, tridiagonal solves along y (k_tridiag_y, + its 8 B per cell table), DST along x
                    moved = 3 * (3 * 16 + 8) * C_
                    rows[k]["kernels"] = "3 solves: DST along x (k_dst_rows*, sources formed in the pass), tridiagonal solves along y (k_tridiag_y), DST along x"
                    rows[k]["bytes_this_build_has_to_move"] = moved
                    rows[k]["frac_of_bytes_this_build_has_to_move"] = moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                if k == "advance_plasma" and ms > 0:
                    # the push against BOTH roofs: HBM above, fp64 VALU here
                    fl = pmc_fp64("void hps::k_advance_tiled<2, 16, false, false, false") if headline else None
                    if fl:
                        simd_clk = 256 * 4 * 2.4e9 * (ms * 1e-3)       # SIMD-cycles of the launch
                        rows[k].update({"flops": fl["flops"], "achieved_tflops": fl["flops"] / (ms * 1e-3) / 1e12,
                                        "peak_fp64_valu_tflops": FP64_VALU_PEAK_TFLOPS,
                                        "frac_fp64_valu": fl["flops"] / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                        "flops_per_particle": fl["flops"] / (args.ppc * args.ppc * C_),
                                        # issue slots: a wave instruction holds its SIMD's 16 lanes for 4 cycles
                                        "frac_simd_issue_fp64": 4.0 * fl["fp64_wave_instructions"] / simd_clk,
                                        "frac_simd_issue_all_valu": 4.0 * fl["valu_wave_instructions"] / simd_clk,
                                        "flops_source": f"profiles/{PMC_FP64}: SQ_INSTS_VALU_ADD/MUL/FMA/TRANS_F64 per launch x 64 lanes, FMA = 2 "
                                                        "(committed rocprofv3 --pmc summary, not collected in this run)"})
            out["roofline"]["per_kernel"] = rows
            out["roofline"]["dominant_by_time"] = max(rows, key=lambda k: rows[k]["us_per_slice"])
            out["roofline"]["furthest_below_the_roofline"] = min((k for k in rows if rows[k]["frac"]), key=lambda k: rows[k]["frac"])
            out["roofline"]["per_kernel_note"] = ("`frac` (top level) stays the deposition's, the kernel the north star prices; per_kernel has every phase: "
                                                  "time from HIP events around the phase"
                                                  + (f" in a window of {phase_window['slices']} slices behind the timed region (slices {phase_window['first']}-{phase_window['last']})"
                                                     if (short and phase_window is not None) else " on the profiled slices of the timed region")
                                                  + ", algorithmic bytes of SURVEY 8(d) (multigrid at the measured V-cycle count), counter bytes from profiles/" + PMC_SUMMARY)
        if args.config5:
            c5 = config5_rows(args.si, args.laser_solver, args.n, args.ppc * args.ppc)
            if c5:
                rows5, src5 = c5
                out["roofline"]["per_kernel"] = rows5
                out["roofline"]["per_kernel_source"] = src5
                out["roofline"]["dominant_by_time"] = max((k for k in rows5 if k != "unlisted_kernels"), key=lambda k: rows5[k]["us_per_slice"])
                out["roofline"]["per_kernel_note"] = ("kernel time per slice by group, from the committed kernel trace of this command; the envelope solve runs on its own "
                                                      "stream beside the slice's chain (HPS_LASER_ASYNC), so the rows sum to more than 1 / value; bytes: SURVEY 8(d)'s "
                                                      "per-particle figures with the sixth gathered plane (|a|^2), electrons 4 per cell, dopant 1 per cell")
        if not (args.config5 or args.config2):
            # the whole slice against the HBM roofline: bytes of the reference's pass structure and of the survey's
            # fused lower bound at the measured V-cycle count, over the measured time per slice of ONE stage
            nv = out["vcycles_per_slice"]
            b_ref, b_fused = slice_bytes(args.n, args.ppc * args.ppc, nv)
            t_slice = 1.0 / out["value"] * world if out["value"] > 0 else 0.0
            cb = pmc_slice_bytes() if headline else None
            out["roofline"]["slice"] = {
                "vcycles": nv,
                # what the algorithm has to move at least (fused lower bound) and what the counters say the engine
                # moves, each over the measured time per slice of ONE stage
                "algorithmic_bytes_fused_lower_bound": b_fused,
                "achieved_fused": b_fused / t_slice / 1e9 if t_slice else 0.0,
                "frac_fused_lower_bound": b_fused / t_slice / 1e9 / HBM_PEAK_GBS if t_slice else 0.0,
                "slices_per_s_at_peak_fused": HBM_PEAK_GBS * 1e9 / b_fused,
                "counter_bytes_per_slice": cb,
                "achieved_counter": cb / t_slice / 1e9 if (cb and t_slice) else None,
                "frac_counter_bytes": cb / t_slice / 1e9 / HBM_PEAK_GBS if (cb and t_slice) else None,
                "counter_source": f"profiles/{PMC_SUMMARY}: sum over all kernels of launches x (FETCH_SIZE x2 + WRITE_SIZE) / slices of that run "
                                  "(committed summary, not collected in this run)" if cb else None,
                # for reference only: the bytes the REFERENCE's pass structure would move (the engine does not move them)
                "reference_passes": {"algorithmic_bytes": b_ref, "achieved": b_ref / t_slice / 1e9 if t_slice else 0.0,
                                     "frac": b_ref / t_slice / 1e9 / HBM_PEAK_GBS if t_slice else 0.0},
                "note": "seconds per slice of one stage = 1 / value per GPU"}
        if args.cpu_slices > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n, args.ppc, args.cpu_slices, args.cpu_threads or min(os.cpu_count() or 1, CPU_THREADS_DEFAULT))
        emit(out)
    if world > 1:
        dist.barrier()
        if transport is not None:
            transport.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
