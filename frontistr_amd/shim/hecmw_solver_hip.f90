!> Reference-side binding of libfistr_hip: a drop-in replacement for
!> hecmw1/src/solver/hecmw_solver.f90 (module hecmw_solver, subroutine hecmw_solve,
!> hecmw_solver.f90:9) -- same module name, same procedure name, same arguments, so that
!> fistr1 (fistr1/src/lib/solve_LINEQ.f90:22 `call hecmw_solve(hecMESH,hecMAT)`) drives the
!> MI355X path unchanged.  Compile it in place of hecmw_solver.f90 in the reference's own
!> source list (hecmw1/src/solver/CMakeLists.txt:20-22) and link -lfistr_hip.
!>
!> The shim only takes c_loc() of the members of hecmwST_matrix / hecmwST_local_mesh and
!> forwards them over the C ABI of include/fistr_hip.h; messages and abort policy follow
!> hecmw_solve_error (hecmw1/src/solver/init/hecmw_solve_error.f90:19-83).
!>
!> HECMW_GPU=0 in the environment keeps the original CPU path (hecmw_solve_iterative); configurations outside the
!> GPU path abort unless HECMW_GPU_UNSUPPORTED=reference asks for the reference's CPU solver for them.
module hecmw_solver
  use iso_c_binding
  implicit none

  type, bind(C) :: fx_matrix_view
    integer(c_int32_t) :: N, NP, NPL, NPU, NDOF
    type(c_ptr) :: indexL, itemL, indexU, itemU
    type(c_ptr) :: D, AL, AU, B, X
  end type fx_matrix_view

  type, bind(C) :: fx_comm_view
    integer(c_int32_t) :: my_rank, PETOT, nn_internal, n_node, n_neighbor_pe
    type(c_ptr) :: neighbor_pe, import_index, import_item, export_index, export_item
  end type fx_comm_view

  type, bind(C) :: fx_solve_info
    integer(c_int32_t) :: iterations, method, precond, ncolor, n_hist
    real(c_double) :: resid, rel_resid, time_setup, time_sol, time_comm, time_matvec, time_precond
  end type fx_solve_info

  interface
    integer(c_int) function fx_create(device, ctx) bind(C, name='fx_create')
      import :: c_int, c_ptr
      integer(c_int), value :: device
      type(c_ptr) :: ctx
    end function fx_create
    integer(c_int) function fx_solve(ctx, mat, comm, Iarray, Rarray, info, hist, hist_len) bind(C, name='fx_solve')
      import :: c_int, c_ptr, c_int32_t, c_double, fx_matrix_view, fx_comm_view, fx_solve_info
      type(c_ptr), value :: ctx
      type(fx_matrix_view) :: mat
      type(fx_comm_view) :: comm
      integer(c_int32_t) :: Iarray(100)
      real(c_double) :: Rarray(100)
      type(fx_solve_info) :: info
      real(c_double) :: hist(*)
      integer(c_int32_t), value :: hist_len
    end function fx_solve
    function fx_last_error() bind(C, name='fx_last_error') result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function fx_last_error
  end interface

  type(c_ptr), save, private :: fx_ctx = c_null_ptr

contains

  subroutine hecmw_solve (hecMESH, hecMAT)
    use hecmw_util
    use hecmw_matrix_misc
    use hecmw_solver_iterative
    use m_hecmw_solve_error
    implicit none
    type (hecmwST_matrix), target :: hecMAT
    type (hecmwST_local_mesh), target :: hecMESH

    type(fx_matrix_view) :: mv
    type(fx_comm_view)   :: cv
    type(fx_solve_info)  :: info
    real(kind=kreal), allocatable, target :: hist(:)
    integer(kind=kint) :: ierr, i, nhist, precond
    character(len=8) :: env
    character(len=16) :: envu
    logical :: on_gpu
    integer :: elen, estat

    ! Explicit opt-out: HECMW_GPU=0 keeps the reference's own CPU solver for every call, and says so.
    ! Configurations outside the GPU hot path -- other block sizes, direct solvers, MPC / contact matrices,
    ! preconditioners other than SSOR / DIAG / ILU(0) -- are REFUSED (E-message + abort), unless the user has asked for
    ! the reference's CPU code for exactly those with HECMW_GPU_UNSUPPORTED=reference.  Nothing is routed silently and
    ! nothing is routed by default.
    call get_environment_variable('HECMW_GPU', env, elen, estat)
    precond = hecMAT%Iarray(3)
    if (estat == 0 .and. elen > 0 .and. env(1:1) == '0') then
      if (hecMESH%my_rank == 0) write(*,'(a)') '### libfistr_hip: reference CPU solver used (HECMW_GPU=0)'
      call hecmw_solve_iterative(hecMESH, hecMAT)
      return
    endif
    on_gpu = hecMAT%Iarray(99) == 1 .and. hecMESH%mpc%n_mpc == 0 .and. hecMAT%cmat%n_val == 0
    if (hecMAT%NDOF == 3) then
      on_gpu = on_gpu .and. (precond == 1 .or. precond == 2 .or. precond == 3 .or. precond == 10)
    else   ! generic block sizes: METHOD 1-4 with SSOR / DIAG (include/fistr_hip.h)
      on_gpu = on_gpu .and. hecMAT%NDOF >= 1 .and. hecMAT%NDOF <= 6 .and. (precond >= 1 .and. precond <= 3) .and. &
               (hecMAT%Iarray(2) >= 1 .and. hecMAT%Iarray(2) <= 4)
    endif
    if (.not. on_gpu) then
      call get_environment_variable('HECMW_GPU_UNSUPPORTED', envu, elen, estat)
      if (estat == 0 .and. elen >= 9 .and. envu(1:9) == 'reference') then
        if (hecMESH%my_rank == 0) write(*,'(a,i0,a,i0,a)') '### libfistr_hip: reference CPU solver used (NDOF=', &
          hecMAT%NDOF, ', PRECOND=', precond, ', MPC / contact / direct: HECMW_GPU_UNSUPPORTED=reference)'
        call hecmw_solve_iterative(hecMESH, hecMAT)
        return
      endif
      if (hecMESH%my_rank == 0) write(*,'(a,i0,a,i0,a,i0,a)') '#### libfistr_hip-E: not on the GPU path (NDOF=', hecMAT%NDOF, &
        ', PRECOND=', precond, ', solver type=', hecMAT%Iarray(99), &
        ', MPC / contact); set HECMW_GPU_UNSUPPORTED=reference to run these on the reference CPU solver'
      call hecmw_abort(hecmw_comm_get_comm())
    endif

    if (.not. c_associated(fx_ctx)) then
      ierr = fx_create(-1_c_int, fx_ctx)        ! device = LOCAL_RANK
      if (ierr /= 0) then
        write(*,*) '#### libfistr_hip: cannot create a device context'
        call hecmw_abort(hecmw_comm_get_comm())
      endif
    endif

    mv%N = hecMAT%N; mv%NP = hecMAT%NP; mv%NPL = hecMAT%NPL; mv%NPU = hecMAT%NPU; mv%NDOF = hecMAT%NDOF
    mv%indexL = c_loc(hecMAT%indexL(0)); mv%itemL = c_loc(hecMAT%itemL(1))
    mv%indexU = c_loc(hecMAT%indexU(0)); mv%itemU = c_loc(hecMAT%itemU(1))
    mv%D = c_loc(hecMAT%D(1)); mv%AL = c_loc(hecMAT%AL(1)); mv%AU = c_loc(hecMAT%AU(1))
    mv%B = c_loc(hecMAT%B(1)); mv%X = c_loc(hecMAT%X(1))

    cv%my_rank = hecMESH%my_rank; cv%PETOT = hecMESH%PETOT
    cv%nn_internal = hecMESH%nn_internal; cv%n_node = hecMESH%n_node
    cv%n_neighbor_pe = hecMESH%n_neighbor_pe
    cv%neighbor_pe = c_null_ptr; cv%import_index = c_null_ptr; cv%import_item = c_null_ptr
    cv%export_index = c_null_ptr; cv%export_item = c_null_ptr
    if (hecMESH%n_neighbor_pe > 0) then
      cv%neighbor_pe  = c_loc(hecMESH%neighbor_pe(1))
      cv%import_index = c_loc(hecMESH%import_index(0)); cv%import_item = c_loc(hecMESH%import_item(1))
      cv%export_index = c_loc(hecMESH%export_index(0)); cv%export_item = c_loc(hecMESH%export_item(1))
    endif

    nhist = max(hecmw_mat_get_iter(hecMAT), 1) + 1   ! GMRES logs MAXIT+1 lines when it runs out
    allocate(hist(nhist))
    ierr = fx_solve(fx_ctx, mv, cv, hecMAT%Iarray, hecMAT%Rarray, info, hist, int(nhist, c_int32_t))

    ! the reference's stdout channel (hecmw_solver_Iterative.f90:418-419, hecmw_solver_CG.f90:245, :168)
    if (hecMESH%my_rank == 0 .and. (hecMAT%Iarray(21) == 1 .or. hecMAT%Iarray(22) >= 1)) then
      write(*,'(a,i0,a,i0,a,i0,a,i0,a,i0)') '### ', hecMAT%NDOF, 'x', hecMAT%NDOF, ' BLOCK (libfistr_hip) METHOD ', &
        info%method, ', PRECOND ', info%precond, ', ', hecMAT%Iarray(5)
    endif
    if (hecMESH%my_rank == 0 .and. hecMAT%Iarray(21) == 1) then
      do i = 1, info%n_hist
        write(*,'(i7, 1pe16.6)') i, hist(i)
      enddo
    endif
    if (hecMESH%my_rank == 0 .and. (hecMAT%Iarray(21) == 1 .or. hecMAT%Iarray(22) >= 1)) then
      write(*,"(a,1pe12.5)") '### Relative residual =', info%rel_resid
    endif
    deallocate(hist)

    if (ierr < 0) then
      write(*,*) '#### libfistr_hip runtime failure'
      call hecmw_abort(hecmw_comm_get_comm())
    else if (ierr /= 0) then
      call hecmw_solve_error(hecMESH, ierr)     ! E-codes abort, W-codes warn (hecmw_solve_error.f90)
    endif
  end subroutine hecmw_solve

  !> The second public procedure of the reference's module (hecmw_solver.f90:80-122, no caller in fistr1): solve hecMATorig as a
  !> system with NDOF unknowns per node.  Kept so that the module interface is complete; the solve itself is hecmw_solve above.
  subroutine hecmw_substitute_solver(hecMESH, hecMATorig, NDOF)
    use hecmw_util
    use hecmw_matrix_contact
    implicit none
    type (hecmwST_local_mesh)      :: hecMESH
    type (hecmwST_matrix)          :: hecMATorig
    integer(kind=kint)             :: NDOF
    type (hecmwST_matrix), pointer :: work
    work => null()
    if (NDOF < hecMATorig%NDOF) call hecmw_abort(hecmw_comm_get_comm())
    if (NDOF == hecMATorig%NDOF) then
      call hecmw_clone_matrix(hecMATorig, work)
    else
      call hecmw_blockmatrix_expand(hecMATorig, work, NDOF)
      call hecmw_cmat_init(work%cmat)
    endif
    call hecmw_solve(hecMESH, work)
    if (NDOF /= hecMATorig%NDOF) call hecmw_vector_contract(hecMATorig, work, NDOF)
  end subroutine hecmw_substitute_solver

end module hecmw_solver
