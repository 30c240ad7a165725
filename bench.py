#!/usr/bin/env python3
"""bench.py -- CG iterations/sec + SpMV achieved HBM GB/s on the 10M-DOF 3x3-block hex mesh
(BASELINE.json metric; configs[2]: synthetic 10M-DOF linear-elastic hex mesh, CG + SSOR(1), fp64).

One "step" = one preconditioned-CG iteration of hecmw_solve_CG (precond apply, 3 dots,
SpMV, 3 AXPYs) on the device-resident system.  Everything is resident in HBM before the
timed region: the matrix is assembled ON the device (fx_assemble_c3d8) from the synthetic
mesh, the multicolour SSOR is set up, r0/||b|| are computed; then W untimed + exactly K timed
iterations run between barrier + synchronize pairs.

N > 1: one process per GPU.  Either the driver starts the ranks (torch.distributed.run: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or `python bench.py --gpus N` starts them
itself: the parent -- which never imports torch and never touches a GPU -- spawns N rank
processes with that environment, relays rank 0's JSON line and fails if any rank fails.
The global cube is split into px*py*pz node-based overlapping subdomains of 150^3 internal nodes
each (weak scaling, the reference's own decomposition model); the SpMV halo exchange and the
dot-product all-reduces go through RCCL inside the library (fx_comm_init); torch.distributed
(gloo) is only the control plane: rendezvous, the ncclUniqueId broadcast, barriers, the max over
ranks of the timed region.  `n_gpus` is what the RCCL communicator reports (ncclCommCount).

The JSON line also carries
  roofline     : the dominant kernel (BELL-64 SpMV, in the variant the timed loop launches: with
                 the fused p.q partial for CG) timed live with HIP events on the solver stream;
                 achieved = algorithmic bytes (72*nb + 4*(nb-N) + 8*(N+1) + 48*N) /
                 time, peak 8 TB/s (MI355X_MICROARCH.md)
  cpu_baseline : the REAL reference (oracle/_ref/ref_solve_omp_o3, HEC-MW compiled from
                 /root/reference with flang -O3 -fopenmp) timed on this box's host cores (all
                 cores of the affinity mask, capped by the cgroup CPU quota) on the SAME full-size system for a bounded
                 number of iterations (40: ~6 s + 3 s of its set-up; rank 0, N=1 only); --cpu-sample-n 69 times a
                 1.03M-DOF sample instead and scales by the DOF ratio.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def decomposition(n):
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(n) or (n, 1, 1)


def spmv_algorithmic_bytes(N, nb):
    # values + column ids + row index (both halves) + x once + y once
    return 72 * nb + 4 * (nb - N) + 2 * 4 * (N + 1) + 24 * N + 24 * N


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--elems", dest="n", type=int, default=149, help="elements per edge per GPU (149 -> 150^3 nodes = 10.125M DOF)")
    ap.add_argument("--precond", type=int, default=1, help="1 SSOR (config 3), 3 block-Jacobi (config 2), 10 ILU(0)")
    ap.add_argument("--method", type=int, default=1, help="1 CG, 2 BiCGSTAB")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=0,
                    help="elements per edge of the cube the reference is timed on; 0 (default) = the full workload (measured, not "
                         "extrapolated: ~30 s at 10.1M DOF incl. the 7 GB hand-over through /dev/shm); e.g. 69 = the 1.03M-DOF sample")
    ap.add_argument("--cpu-sample-iters", type=int, default=40)
    ap.add_argument("--standard", action="store_true",
                    help="headline = hecmw_solve_CG's loop as written (FX_EISENSTAT=0).  Default: the library's default path, which for CG + "
                         "multicolour SSOR is Eisenstat's one-pass form (same iterates to rounding, matrix streamed once per iteration); the "
                         "other recurrence is reported beside it under `variants`")
    ap.add_argument("--eisenstat", action="store_true", help="(accepted for compatibility: Eisenstat's form is the default since round 4)")
    ap.add_argument("--cpu-full", action="store_true", help="(default since round 3) time the reference on the full workload")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher.  The parent stays free of torch /
# HIP; the children are fresh processes (no exec of a process that has touched the GPU).
# ---------------------------------------------------------------------------------------------
def launch_ranks(a):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed = None
    out0 = b""
    deadline = time.time() + float(os.environ.get("FX_BENCH_LAUNCH_TIMEOUT", "3000"))
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = (bad[0], codes[bad[0]])
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed = (-1, "timeout")
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:           # the exact processes started above, nothing else
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    out0 = procs[0].stdout.read() if procs[0].stdout else b""
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    if failed is not None:
        sys.stderr.write("bench.py: rank %s failed (%s); no result\n" % failed)
        return 1
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    print(lines[-1], flush=True)
    return 0


def check_ledgers(leds):
    """Cross-rank consistency of the communication ledgers (SolverContext.comm_ledger of every rank, rank order)."""
    errors = []
    ref = leds[0]
    for r, g in enumerate(leds):
        for k in ("ops", "seq_hash", "allreduces", "allreduce_bytes", "halos"):
            if g[k] != ref[k]:
                errors.append("rank %d %s = %r, rank 0 has %r" % (r, k, g[k], ref[k]))
        for p, (ns, bs, nr, br) in g["peers"].items():
            if not 0 <= p < len(leds):
                errors.append("rank %d lists neighbour %d outside the world" % (r, p))
                continue
            back = leds[p]["peers"].get(r)
            if back is None:
                errors.append("rank %d exchanges with %d, which does not list it" % (r, p))
            elif (ns, bs) != (back[2], back[3]) or (nr, br) != (back[0], back[1]):
                errors.append("rank %d -> %d: %d messages / %d bytes sent, %d / %d received there; %d <- %d: %d / %d received, %d / %d sent there"
                              % (r, p, ns, bs, back[2], back[3], r, p, nr, br, back[0], back[1]))
    return {"errors": errors, "ops_per_rank": ref["ops"], "allreduces": ref["allreduces"], "halo_exchanges": ref["halos"],
            "allreduce_bytes": ref["allreduce_bytes"], "own_halo_comm": ref["own_halo_comm"],
            "halo_bytes_sent_per_rank": [sum(v[1] for v in g["peers"].values()) for g in leds]}


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (on the GPU boxes of this
    pool: 256 cores in the mask, cpu.max = 16 CPUs; 256 OpenMP threads on a 16-CPU quota ran the reference 12x slower)."""
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_baseline(hip, np, n_sample, iters, cores, method=1, precond=1):
    """Reference CG + multicolour SSOR on a (n_sample+1)^3-node cube, `iters` iterations."""
    from frontistr_amd.mesh import CubeMesh
    from oracle import refrun
    exe = "ref_solve_omp_o3" if refrun.have_ref("ref_solve_omp_o3") else "ref_solve_omp"
    if not refrun.have_ref(exe):
        return None
    mesh = CubeMesh(n_sample)
    hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
    hm.elem_node_item = mesh.conn.ravel()
    m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    ctx.download_matrix(m)          # the sample system is exactly what the GPU path assembled
    ctx.close()
    A = refrun.BSR(m.N, m.NP, m.indexL, m.itemL, m.indexU, m.itemU, m.D, m.AL, m.AU, m.B)
    # ITERLOG on: the reference's residual history of these iterations ('(i7,1pe16.6)', hecmw_solver_CG.f90:245) is the full-size
    # parity evidence of the JSON line (parity_fullsize); 40 printed lines cost nothing next to 40 iterations of 0.14 s
    I, R = refrun.default_params(method=method, precond=precond, maxit=iters, tol=1e-30, iterlog=1, timelog=1)
    wd = "/dev/shm" if os.path.isdir("/dev/shm") else None
    r = refrun.run_solve(A, I, R, threads=cores, workdir=wd, timeout=2400, exe_name=exe)
    if "t_per_iter" not in r or r["t_per_iter"] <= 0:
        return None
    return dict(per_iter=r["t_per_iter"], solver=r.get("t_solver"), setup=r.get("t_setup"),
                matvec=r.get("t_matvec"), precond=r.get("t_precond"), ndof=3 * mesh.n_node, exe=exe,
                history=[h for _, h in r.get("history", [])])


def main():
    a = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(launch_ranks(a))
    world = int(env_world or "1")
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; they must agree" % (a.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # Native libraries write to the process' stdout (RCCL prints a version banner when a communicator is created, the
    # reference binaries of the cpu_baseline leg log their solver summary): keep fd 1 for the ONE JSON line only.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from frontistr_amd import hecmw as hip
    from frontistr_amd.mesh import CubeMesh

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libfistr_hip has no CPU path")
    # FX_BENCH_TRANSPORT=gloo: rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices,
    # halos / reductions go through the library's host-callback transport).  Default: RCCL over xGMI.
    transport = os.environ.get("FX_BENCH_TRANSPORT", "rccl")
    ndev = torch.cuda.device_count()
    if world > 1 and transport != "gloo" and ndev < world:
        raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, this box has %d "
                         "(FX_BENCH_TRANSPORT=gloo rehearses the decomposition with ranks sharing devices)" % (world, world, ndev))
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo")     # control plane only; the data path is RCCL inside libfistr_hip

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    t_setup0 = time.time()
    ctx = hip.SolverContext(device=dev)
    E, NU = 210000.0, 0.3
    if world == 1:
        mesh = CubeMesh(a.n)
        hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
        coord, conn, load, bc = mesh.coord, mesh.conn, mesh.load(), mesh.dirichlet()
        if os.environ.get("FX_FORCE_COMM", "0") not in ("", "0"):
            # rehearsal of the in-stream RCCL reductions on one GPU: a 1-rank communicator, every scalar stage goes
            # reduce -> ncclAllReduce -> logic exactly as in a multi-GPU run (no halo: there are no neighbours)
            ctx.comm_init(hip.comm_unique_id(), 0, 1)
    else:
        from frontistr_amd.partition import cube_subdomain
        sub = cube_subdomain(a.n + 1, decomposition(world), rank)
        hm = sub.hecmesh(hip)
        if transport == "gloo":
            from frontistr_amd.comm import attach_host_callbacks
            attach_host_callbacks(ctx, hm, hip.lib())
        else:
            uid = torch.tensor(list(hip.comm_unique_id()) if rank == 0 else [0] * 128, dtype=torch.uint8)
            dist.broadcast(uid, 0)
            ctx.comm_init(bytes(uid.tolist()), rank, world)
        coord, conn, load, bc = sub.coord, sub.conn, sub.load(), sub.dirichlet()
    comm_ranks, comm_dev = ctx.comm_size()
    if comm_ranks != world:
        raise SystemExit("bench.py: the transport reports %d ranks, the launcher started %d" % (comm_ranks, world))
    hm.elem_node_item = conn.ravel()
    t0 = time.time()
    m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
    t_con = time.time() - t0               # hecmw_mat_con alone (round 1 also counted context creation and mesh generation here)
    t0 = time.time()
    ctx.upload(m, hm, what=hip.FX_UP_PROFILE)
    t_upload = time.time() - t0            # profile to the device; a large system's value arena is taken here (hipMalloc of fresh VRAM: ~27 ms per GiB)
    ms_asm_first = ctx.assemble_c3d8(coord, conn, E, NU, elemopt=1, load=load, bc=bc)   # builds the element colouring + scatter map
    ms_asm = ctx.assemble_c3d8(coord, conn, E, NU, elemopt=1, load=load, bc=bc)          # what a Newton iteration pays
    m.Iarray[0] = a.warmup + a.steps + 8          # MAXIT: never reached inside the timed region
    m.Iarray[1] = a.method
    m.Iarray[2] = a.precond
    m.Rarray[0] = 1.0e-30                          # tolerance far below reach: no early exit
    t0 = time.time()
    two_forms = (a.method == 1 and a.precond == 1 and os.environ.get("FX_EISENSTAT", "1") not in ("0",))
    ctx.precond_setup(m)                    # library defaults (on a subdomain the set-up also builds the halo-column layout the one-pass form needs)
    if two_forms and a.standard:
        ctx.set_option("FX_EISENSTAT", 0)   # --standard: the headline loop is the reference's recurrence as written
    t_pre = time.time() - t0
    placement = ctx.placement_report()   # the value arena: nothing is timed or searched at set-up since round 4
    st = ctx.stats()
    N, nb = st["N"], st["M_blocks"]

    # A step that does not come back (ranks that disagree on the sequence of RCCL operations wait for each other for ever) must end
    # as a message, not as a silent hang: the watchdog prints this rank's communication ledger -- how many operations it has issued,
    # the hash of their sequence, messages per neighbour -- and ends the process.  FX_BENCH_WATCHDOG seconds per phase (0 = off).
    import threading
    wd_s = float(os.environ.get("FX_BENCH_WATCHDOG", "600"))
    wd_phase = {"name": None, "t0": 0.0}

    def _watch():
        while True:
            time.sleep(1.0)
            nm = wd_phase["name"]
            if nm and wd_s > 0 and time.time() - wd_phase["t0"] > wd_s:
                try:
                    led = ctx.comm_ledger()
                except Exception as e:
                    led = repr(e)
                sys.stderr.write("bench.py: rank %d: phase '%s' did not finish within %.0f s -- communication ledger of this rank: %r\n"
                                 "bench.py: compare `ops` / `seq_hash` across ranks: the first rank that differs issued a different sequence of "
                                 "all-reduces / halo exchanges (DESIGN.md section 6)\n" % (rank, nm, wd_s, led))
                sys.stderr.flush()
                os._exit(3)

    threading.Thread(target=_watch, daemon=True).start()

    def phase(name):
        wd_phase["t0"] = time.time()
        wd_phase["name"] = name

    phase("krylov_begin + warm-up")
    ctx.krylov_begin(m)
    it, status, resid = ctx.krylov_steps(a.warmup)
    barrier()
    phase("timed iterations")
    t0 = time.perf_counter()
    it, status, resid = ctx.krylov_steps(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    phase(None)
    assert status == 0 and it == a.warmup + a.steps + 1, (status, it)
    assert np.isfinite(resid)
    headline_eis = bool(ctx.stats()["eisenstat"])       # the recurrence the timed loop really ran in (read before anything else touches the context)
    hist_gpu = ctx.krylov_history()                     # its ITERLOG lines 1 .. warmup + steps (outside the timed region)
    devices_used = 1
    ledger_check = None
    if world > 1:
        t = torch.tensor([dt, float(dev)], dtype=torch.float64)
        g = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(g, t)
        dt = max(float(x[0]) for x in g)
        devices_used = len(set(int(x[1]) for x in g))
        # every rank's communication ledger over the control plane: same operation count and sequence hash everywhere, and what r
        # sent to p is what p received from r (messages and bytes) -- checked on every rank, so that every rank fails together
        leds = [None] * world
        dist.all_gather_object(leds, ctx.comm_ledger())
        ledger_check = check_ledgers(leds)
        if ledger_check["errors"]:
            raise SystemExit("bench.py: the ranks disagree on their communication: " + "; ".join(ledger_check["errors"][:8]))

    # The OTHER recurrence of CG + SSOR in the same line: same context, data and placement, timed exactly like the headline
    # (W untimed + K timed iterations between barriers, max over ranks).  Bytes of one iteration:
    #   standard : SpMV + SSOR apply + 480 N of vector work
    #   eisenstat: L and U once (values + column ids), the diagonal factors in both sweeps and in the update (D~ x is formed from
    #              them: no second diagonal array), 21 vector passes
    def form_bytes(eis):
        if eis:
            return 76 * (st["L_blocks"] + st["U_blocks"]) + 3 * 72 * N + 21 * 24 * N
        pb = (76 * (st["L_blocks"] + st["U_blocks"]) + 2 * 72 * N + 24 * N + 4 * 24 * N) if a.precond in (1, 10) else (72 * N + 48 * N)
        return spmv_algorithmic_bytes(N, nb) + pb + 480 * N

    variants = {}
    try:
        if two_forms:
            other_eis = not headline_eis
            ctx.set_option("FX_EISENSTAT", 1 if other_eis else 0)
            ctx.krylov_begin(m)
            ctx.krylov_steps(a.warmup)
            barrier()
            t0 = time.perf_counter()
            it_v, status_v, resid_v = ctx.krylov_steps(a.steps)
            barrier()
            dt_v = time.perf_counter() - t0
            active = bool(ctx.stats()["eisenstat"])
            ctx.set_option("FX_EISENSTAT", 1 if headline_eis else 0)
            if world > 1:
                tv = torch.tensor([dt_v], dtype=torch.float64)
                gv = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(gv, tv)
                dt_v = max(float(x[0]) for x in gv)
            if active == other_eis and status_v == 0 and it_v == a.warmup + a.steps + 1:
                vb = form_bytes(other_eis)
                variants["eisenstat" if other_eis else "standard"] = {
                    "what": ("CG + multicolour SSOR(1) in Eisenstat's one-pass form (same iterates to rounding, matrix streamed once per iteration)" if other_eis
                             else "hecmw_solve_CG's loop as written (FX_EISENSTAT=0): SSOR apply + SpMV per iteration, matrix streamed twice"),
                    "it_per_s": world * a.steps / dt_v, "ms_per_step": 1e3 * dt_v / a.steps, "bytes": vb,
                    "achieved_GBs": vb / (dt_v / a.steps) / 1e9, "frac": vb / (dt_v / a.steps) / 1e9 / HBM_PEAK_GBS,
                    "resid_after_steps": resid_v,
                }
    except Exception as e:      # an optional sub-record must never cost the headline line
        variants["variant_error"] = repr(e)
        try:
            ctx.set_option("FX_EISENSTAT", 1 if headline_eis else 0)
        except Exception:
            pass

    # roofline of the dominant kernel, timed live with HIP events on the solver stream: the variant the timed loop
    # launches (CG: SpMV with the fused p.q partial; BiCGSTAB: the plain product)
    spmv_variant = 1 if a.method == 1 else 0
    ms_spmv = ctx.spmv_resident_ms(spmv_variant, 20)
    ms_prec = ctx.precond_apply_ms(10)
    stream_gbs = ctx.stream_ceiling_gbs(5)       # on-box read-streaming ceiling
    alg = spmv_algorithmic_bytes(N, nb)
    achieved = alg / (ms_spmv * 1e-3) / 1e9
    # streamed bytes of one SSOR apply: L and U blocks (values + column ids), ALU twice, r once, z r/w twice
    if a.precond in (1, 10):
        prec_bytes = 76 * (st["L_blocks"] + st["U_blocks"]) + 2 * 72 * N + 24 * N + 4 * 24 * N
    else:
        prec_bytes = 72 * N + 48 * N
    bell_pad = st["M_pairs"] * 64.0 / max(nb, 1) - 1.0   # M_pairs counts block positions
    traffic, traffic_src, asm_traffic, asm_traffic_src = None, None, None, None
    n_elem = int(conn.shape[0])
    try:  # PMC-measured HBM bytes per launch exist only for profiled workloads (committed under profiles/)
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            tjs = json.load(open(path))
            tj = tjs.get("k_spmv", {}).get(str(N))
            if tj and world == 1 and traffic is None:
                traffic, traffic_src = tj["traffic_bytes"], tj["source"]
            ta = tjs.get("k_assemble_c3d8", {}).get(str(n_elem))
            if ta and world == 1 and asm_traffic is None:
                asm_traffic, asm_traffic_src = ta["traffic_bytes"], ta["source"]
    except Exception:
        pass
    # stiffness assembly (fstr_StiffMatrix + hecmw_mat_ass_elem, IC element): every element adds its 64 blocks of 72 bytes into
    # D / AL / AU -- one read and one write of each -- on top of the zeroed 6.9 GB result
    asm_bytes = 2 * 64 * 72 * n_elem

    out = {
        "metric": "CG iterations/sec + SpMV achieved HBM GB/s, 10M-DOF 3x3-block mesh",
        # weak scaling: every rank advances its own 10.125M-DOF subdomain by K iterations, so the job
        # processed world*K subdomain-iterations (N=1: plain CG iterations/s on the 10M-DOF mesh)
        "value": world * a.steps / dt,
        "unit": "CG iterations/s" if world == 1 else "CG iterations/s x subdomains (%.3fM-DOF subdomain-iterations/s, summed over GPUs)" % (3 * N / 1e6),
        "global_iterations_per_s": a.steps / dt,
        # devices really used: equal to the rank count over RCCL (one rank per GPU); in the gloo rehearsal mode ranks SHARE devices,
        # so such a line is not a scaling point (rehearsal_ranks says how many subdomains were time-sliced on those devices)
        "n_gpus": comm_ranks if transport != "gloo" or world == 1 else devices_used,
        "ranks": comm_ranks,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "synthetic %d^3-node linear-elastic C3D8 cube per GPU (%.3fM DOF/GPU, %.2fM DOF total), %s + %s, fp64"
                        % (a.n + 1, 3 * N / 1e6, 3 * N * world / 1e6, {1: "CG", 2: "BiCGSTAB"}[a.method],
                           {1: "SSOR(1) multicolour", 3: "block-Jacobi", 10: "ILU(0) level-scheduled"}[a.precond]),
            "decomposition": "x".join(str(d) for d in decomposition(world)),
            "rehearsal_ranks": comm_ranks if (transport == "gloo" and world > 1) else None,
            "transport": "none" if world == 1 else ("rccl (ncclCommCount=%d)" % comm_ranks if transport != "gloo" else "gloo host callbacks (rehearsal)"),
            "devices_used": devices_used,
            "ncolor": st["ncolor"],
            "recurrence": ("eisenstat one-pass form (library default for CG + multicolour SSOR; FX_EISENSTAT=0 / --standard opts out)" if headline_eis
                           else "standard (hecmw_solve_CG / hecmw_solve_BiCGSTAB as written)"),
            "block_rows": N, "blocks": nb,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_spmv<0,%d> (BELL-64 3x3-block SpMV%s)" % (spmv_variant, ", fused p.q partial: the launch of the timed CG loop" if spmv_variant else ""),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "measured_stream_ceiling": stream_gbs, "frac_of_measured_ceiling": achieved / stream_gbs,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": alg, "ms_per_launch": ms_spmv,
            "bell_padding_frac": bell_pad,
            "precond_apply": {"ms": ms_prec, "algorithmic_bytes": prec_bytes,
                              "achieved_GBs": prec_bytes / (ms_prec * 1e-3) / 1e9,
                              "frac": prec_bytes / (ms_prec * 1e-3) / 1e9 / HBM_PEAK_GBS},
            # the whole timed iteration in the recurrence it ran in: algorithmic bytes of one iteration / measured time per iteration
            "iteration": {"recurrence": "eisenstat" if headline_eis else "standard", "algorithmic_bytes": form_bytes(headline_eis),
                          "ms": 1e3 * dt / a.steps, "achieved_GBs": form_bytes(headline_eis) / (dt / a.steps) / 1e9,
                          "frac": form_bytes(headline_eis) / (dt / a.steps) / 1e9 / HBM_PEAK_GBS},
            "iteration_GBs": form_bytes(headline_eis) / (dt / a.steps) / 1e9,
            "assembly": {"kernel": "k_assemble_c3d8<IC> + load + k_bc_apply (fx_assemble_c3d8, second call: colouring and scatter map built)",
                         "ms": ms_asm, "ms_first_call": ms_asm_first, "algorithmic_bytes": asm_bytes,
                         "achieved_GBs": asm_bytes / (ms_asm * 1e-3) / 1e9, "frac": asm_bytes / (ms_asm * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": asm_traffic, "traffic_source": asm_traffic_src},
            # where the value arrays live: ONE arena taken before anything else of the system, every array at a fixed offset in it
            # (what makes the speed class of this kernel the same in every process)
            "placement": placement,
        },
        "setup_s": {"mat_con": t_con, "profile_upload_and_arena": t_upload, "assemble_ms": ms_asm, "precond_setup": t_pre,
                    "note": "precond_setup = ordering + colouring + layouts + factors (no placement search, no timing of candidates: the value arena "
                            "is taken at fx_upload, before the CSR arrays)"},
        "resid_after_steps": resid,
        "variants": variants,
    }
    if ledger_check is not None:
        # what the ranks exchanged up to the end of the timed iterations (set-up + warm-up + K iterations), cross-checked rank against rank
        out["comm_ledger"] = {k: v for k, v in ledger_check.items() if k != "errors"}
        out["comm_ledger"]["consistent"] = True

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cores = host_cores()     # every core this process may run on (affinity mask capped by the cgroup quota)
        n_cpu = a.n if (a.cpu_full or a.cpu_sample_n <= 0 or a.cpu_sample_n >= a.n) else a.cpu_sample_n
        try:
            cb = cpu_baseline(hip, np, n_cpu, a.cpu_sample_iters, cores, a.method, a.precond)
        except Exception as e:  # the baseline is reporting only; never fail the bench on it
            cb = None
            out["cpu_baseline_error"] = repr(e)
        if cb:
            scale = cb["ndof"] / (3.0 * N)      # memory-bound sweeps: time per iteration ~ DOF
            extrap = abs(scale - 1.0) > 1e-12
            out["cpu_baseline"] = {
                "value": (1.0 / cb["per_iter"]) * scale, "unit": "CG iterations/s", "cores": cores,
                "kind": "reference", "extrapolated": extrap,
                "sample": "HEC-MW reference (%s: flang %s -fopenmp, OMP_NUM_THREADS=%d = every core of the affinity mask / cgroup quota), same METHOD/PRECOND "
                          "as the GPU run, on a %d^3-node cube (%.2fM DOF), %d iterations: %.4f s/iter measured = %.2f it/s%s"
                          % (cb["exe"], "-O3" if cb["exe"].endswith("_o3") else "-O2", cores, n_cpu + 1, cb["ndof"] / 1e6,
                             a.cpu_sample_iters, cb["per_iter"], 1.0 / cb["per_iter"],
                             "; value = that rate scaled by DOF ratio %.4f to the %.2fM-DOF workload" % (scale, 3 * N / 1e6)
                             if extrap else " (the full workload: measured, not extrapolated)"),
                "measured_it_per_s_at_sample": 1.0 / cb["per_iter"],
                "sample_matvec_s": cb.get("matvec"), "sample_precond_s": cb.get("precond"),
                "sample_setup_s": cb.get("setup"),
            }
            # Full-size parity (VERDICT r03 #5a): the reference ran the IDENTICAL system (the matrix the GPU assembled, downloaded) from the
            # same X0 = 0; its printed residual history against the GPU loop's own lines of the warm-up + timed iterations.  7 printed
            # digits on the reference's side; the GPU sums in a different, fixed order.
            href = np.array(cb.get("history") or [], dtype=np.float64)
            nl = int(min(len(href), len(hist_gpu))) if not extrap else 0
            if nl > 0:
                rel = np.abs(hist_gpu[:nl] - href[:nl]) / href[:nl]
                out["parity_fullsize"] = {
                    "lines": nl, "max_rel_diff": float(rel.max()), "max_rel_diff_first10": float(rel[:min(10, nl)].max()),
                    "recurrence_gpu": "eisenstat" if headline_eis else "standard", "reference": cb["exe"],
                    "resid_line_%d" % nl: {"gpu": float(hist_gpu[nl - 1]), "reference": float(href[nl - 1])},
                    "note": "RESID = ||r||/||b|| per iteration, GPU loop vs the reference's ITERLOG lines on the identical %.2fM-DOF system; "
                            "the reference prints 7 digits" % (3 * N / 1e6)}
    if rank == 0:
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)                      # whatever teardown prints goes to stderr again
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
