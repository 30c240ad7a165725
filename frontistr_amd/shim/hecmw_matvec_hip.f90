!> Reference-side binding of libfistr_hip for `hecmw_matvec(hecMESH, hecMAT, X, Y, COMMtime)`
!> (module hecmw_solver_las, hecmw1/src/solver/las/hecmw_solver_las.f90:57-77; external callers: implicit dynamics
!> fistr1/src/analysis/dynamic/transit/fstr_dynamic_nlimplicit.f90:155,157,574,576, eigen output fstr_EIG_output.f90:101).
!>
!> A maintainer adds four lines to hecmw_solver_las.f90 (INTEGRATION.md section 2; oracle/build_ref.py applies exactly that
!> patch to a scratch copy when it builds oracle/_ref/shim_solve, so the binding below is compiled against the reference's
!> own .mod files, linked with its objects and run on the GPU by tests/test_gpu_fortran_shim.py):
!>
!>     use hecmw_matvec_hip                                    ! next to the other `use` lines of the module
!>     ...
!>     if (hecmw_matvec_hip_enabled(hecMAT)) then              ! first statements of subroutine hecmw_matvec
!>       call hecmw_matvec_on_gpu(hecMESH, hecMAT, X, Y, COMMtime); return
!>     endif
!>
!> Opt-in (HECMW_GPU_MATVEC=1): hecmw_matvec carries no "matrix changed" flag, so the values of hecMAT are uploaded on
!> every call (6.5 GB at 10M DOF, ~0.1 s) -- right for the external callers, who change hecMAT between products, wrong
!> for the reference's own CPU Krylov loops, which call it once per iteration.  HECMW_GPU_MATVEC=resident says the
!> matrix does not change between calls: values go up on the first call only.
module hecmw_matvec_hip
  use iso_c_binding
  use hecmw_util
  use hecmw_hip_binding
  implicit none
  private
  public :: hecmw_matvec_hip_enabled, hecmw_matvec_on_gpu
  integer, save :: mode = -1          ! -1 not read yet, 0 off, 1 upload every call, 2 resident after the first call

contains

  logical function hecmw_matvec_hip_enabled(hecMAT)
    type(hecmwST_matrix), intent(in) :: hecMAT
    character(len=16) :: env
    integer :: elen, estat
    if (mode < 0) then
      mode = 0
      call get_environment_variable('HECMW_GPU_MATVEC', env, elen, estat)
      if (estat == 0 .and. elen >= 1) then
        if (env(1:1) == '1') mode = 1
        if (elen >= 8) then
          if (env(1:8) == 'resident') mode = 2
        endif
      endif
    endif
    hecmw_matvec_hip_enabled = mode > 0 .and. hecMAT%NDOF >= 1 .and. hecMAT%NDOF <= 6 .and. hecMAT%cmat%n_val == 0
  end function hecmw_matvec_hip_enabled

  !> Y(1:NDOF*N) = (D + AL + AU) X after the halo update of X (whose halo part is written, as hecmw_update_3_R does
  !> inside hecmw_matvec_33_inner, hecmw_solver_las_33.f90:242-246).
  subroutine hecmw_matvec_on_gpu(hecMESH, hecMAT, X, Y, COMMtime)
    type(hecmwST_local_mesh), intent(in), target :: hecMESH
    type(hecmwST_matrix), intent(in), target :: hecMAT
    real(kind=kreal) :: X(:)
    real(kind=kreal) :: Y(:)
    real(kind=kreal), intent(inout), optional :: COMMtime
    type(fx_matrix_view) :: mv
    type(fx_comm_view) :: cv
    type(c_ptr) :: ctx
    real(kind=kreal), allocatable :: xs(:), ys(:)
    real(c_double) :: tcomm
    integer(kind=kint) :: nd, n, np
    integer(c_int) :: ierr
    nd = hecMAT%NDOF; n = hecMAT%N; np = hecMAT%NP
    ctx = fxb_context(hecMESH)
    call fxb_ensure_transport(hecMESH, nd)
    call fxb_views(hecMESH, hecMAT, mv, cv)
    mv%B = c_null_ptr; mv%X = c_null_ptr
    ! "the resident values" (include/fistr_hip.h: mat->D == NULL) only while they are still the ones THIS binding uploaded for
    ! THIS matrix: a hecmw_solve in between puts its own (boundary-condition-modified) matrix into the shared context
    if (mode == 2 .and. fxb_values_owner == 2 .and. c_associated(fxb_values_addr, c_loc(hecMAT%D(1)))) then
      mv%D = c_null_ptr; mv%AL = c_null_ptr; mv%AU = c_null_ptr
    endif
    allocate(xs(nd * np), ys(nd * np))        ! contiguous images (X, Y are assumed-shape and may be sections)
    xs(1:nd * np) = X(1:nd * np)
    ys = 0.d0
    tcomm = 0.d0
    ierr = fx_matvec(ctx, mv, cv, xs, ys, tcomm)
    if (ierr /= 0) then
      write(*,'(a,a)') '#### libfistr_hip-E: hecmw_matvec failed: ', trim(fxb_error_text())
      call hecmw_abort(hecmw_comm_get_comm())
    endif
    fxb_values_owner = 2
    fxb_values_addr = c_loc(hecMAT%D(1))
    Y(1:nd * n) = ys(1:nd * n)
    if (np > n) X(nd * n + 1:nd * np) = xs(nd * n + 1:nd * np)
    if (present(COMMtime)) COMMtime = COMMtime + tcomm
    deallocate(xs, ys)
  end subroutine hecmw_matvec_on_gpu

end module hecmw_matvec_hip
