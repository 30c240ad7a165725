"""Worker for the multi-process tests (spawned by test_distributed.py / test_gpu_distributed.py).
argv: mode(oracle|hip) rank world port m method precond outfile"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


NL_MATERIAL = ((1.0e5, 0.3), dict(plastic=True, harden=0, plconst=(1000.0, 2000.0, 0.0), nlgeom_flag=2))


def nl_bc(coord, clamp, ztop):
    """bottom clamped (the deck's FIX group), top face pulled by 2 % in z with a 1 % x shear"""
    top = (np.nonzero(coord[:, 2] == ztop)[0] + 1).astype(np.int32)
    tn = np.repeat(top, 3).astype(np.int32)
    td = np.tile(np.array([1, 2, 3], dtype=np.int32), top.size)
    tv = np.tile(np.array([0.01 * ztop, 0.0, 0.02 * ztop]), top.size)
    return (np.concatenate([clamp[0], tn]), np.concatenate([clamp[1], td]), np.concatenate([clamp[2], tv]))


def main():
    mode, rank, world, port, m, meth, pc, out = sys.argv[1:9]
    rank, world, meth, pc = int(rank), int(world), int(meth), int(pc)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from frontistr_amd.partition import cube_subdomain
    from frontistr_amd.comm import NeighborExchange
    if m.startswith("dist:"):
        # subdomain written by the REFERENCE partitioner (HECMW-DIST file), deck of the synthetic cube:
        # !BOUNDARY FIX,1,3,0.0 and !CLOAD TOP,1,1.0 through the node groups of the file
        from frontistr_amd.hecmw_dist import read_dist
        sub = read_dist("%s.%d" % (m[5:], rank))
        fix = sub.group("FIX")
        bc = (np.repeat(fix, 3).astype(np.int32), np.tile(np.array([1, 2, 3], dtype=np.int32), fix.size),
              np.zeros(3 * fix.size))
        load = np.zeros(3 * sub.n_node)
        load[3 * (sub.group("TOP") - 1)] = 1.0
        sub.dirichlet = lambda: bc
        sub.load = lambda: load
    else:
        dims = {2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}[world]
        sub = cube_subdomain(int(m), dims, rank)
    E, NU = 210000.0, 0.3
    wg = (0, 0, 0)
    if mode == "oracle":
        from oracle import pyoracle as po
        from oracle.refrun import default_params
        A = po.assemble(1, sub.coord, sub.conn, E, NU, bc=sub.dirichlet(), load=sub.load())
        A.N = sub.nn_internal                                    # rows 1..N are solved, N+1..NP are halo
        ex = NeighborExchange(sub.neighbor_pe, sub.import_index, sub.export_index)
        exp0, imp0 = sub.export_item - 1, sub.import_item - 1

        def halo(x):                                             # hecmw_update_3_R on a full vector
            xv = x.reshape(-1, 3)
            send = np.ascontiguousarray(xv[exp0]).ravel()
            recv = np.zeros(3 * ex.n_import)
            ex.exchange(send, recv)
            xv[imp0] = recv.reshape(-1, 3)

        comm = po.Comm(halo=halo, allreduce=ex.allreduce_sum, nvec=3 * sub.n_node)
        I, R = default_params(method=meth, precond=pc)
        o = po.solve_iterative(A, I, R, nthreads=4, comm=comm)
        X, it, hist, code = o["X"], o["iter"], o["history"], o["code"]
    elif mode == "hipnl":
        # nonlinear static loop (elastoplastic, updated Lagrange) on the subdomain: 2 substeps x 4 Newton iterations
        from frontistr_amd import fstr, hecmw as hip
        from frontistr_amd.comm import attach_host_callbacks
        hm = sub.hecmesh(hip)
        hm.elem_node_item = sub.conn.ravel()
        mat = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
        ctx = hip.SolverContext(device=0)
        attach_host_callbacks(ctx, hm, hip.lib())
        ctx.upload(mat, hm, what=hip.FX_UP_PROFILE)
        solid = fstr.fstr_solid(ctx, sub.coord, sub.conn, fstr.tMaterial(*NL_MATERIAL[0], **NL_MATERIAL[1]))
        bc = nl_bc(sub.coord, sub.dirichlet(), dims[2] * int(m) - 1)
        mat.Iarray[0] = 10000; mat.Iarray[1] = meth; mat.Iarray[2] = pc; mat.Rarray[0] = 1e-10
        log = fstr.fstr_solve_NLGEOM(solid, mat, bc, None, 2, 4, 1e-12)
        st = solid.get_state(("unode", "qforce"))
        X, it, hist, code = st["unode"], log.shape[0], log[:, 4], 0
        ctx.close()
    else:
        from frontistr_amd import hecmw as hip
        from frontistr_amd.comm import attach_host_callbacks
        hm = sub.hecmesh(hip)
        hm.elem_node_item = sub.conn.ravel()
        mat = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
        ctx = hip.SolverContext(device=0)
        attach_host_callbacks(ctx, hm, hip.lib())
        ctx.upload(mat, hm, what=hip.FX_UP_PROFILE)
        ctx.assemble_c3d8(sub.coord, sub.conn, E, NU, elemopt=1, load=sub.load(), bc=sub.dirichlet())
        mat.Iarray[0] = 10000; mat.Iarray[1] = meth; mat.Iarray[2] = pc
        code = ctx.solve_resident(mat)
        ctx.download_x(mat)
        X, it, hist = mat.X, ctx.info.iterations, ctx.history
        st = ctx.stats()
        wg = (st["wg_interior"], st["wg_boundary"], st["eisenstat"])
        ctx.close()
    np.savez(out, X=X, it=it, hist=hist, code=code, gid=sub.global_id, nn_internal=sub.nn_internal, wg=np.array(wg))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
