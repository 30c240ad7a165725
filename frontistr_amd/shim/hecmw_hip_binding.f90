!> Reference-side binding of libfistr_hip, common part: the C-ABI declarations of include/fistr_hip.h, the one device
!> context of this MPI rank (the state the reference keeps in module `save` variables), the views of
!> hecmwST_matrix / hecmwST_local_mesh (hecmw1/src/common/hecmw_util_f.F90:433-468, :298-310) and the TRANSPORT of a
!> domain-decomposed run (`mpirun -np N fistr1`):
!>
!>   HECMW_GPU_TRANSPORT=rccl (default)  rank 0 makes the ncclUniqueId (fx_comm_unique_id), hecmw_bcast_C distributes it
!>                                       over hecMESH%MPI_COMM, every rank calls fx_comm_init: halo exchange and dot-product
!>                                       reductions then stay on the device (RCCL over xGMI, no host round trip);
!>   HECMW_GPU_TRANSPORT=mpi             fx_comm_set_host_callbacks wired to the reference's own hecmw_update_m_R
!>                                       (hecmw_comm_f.F90:816-841 -> hecmw_solve_send_recv_mm) and hecmw_allreduce_R
!>                                       (:346-379): any MPI the reference was built with, no RCCL needed.
!>
!> Used by frontistr_amd/shim/hecmw_solver_hip.f90 (module hecmw_solver: hecmw_solve) and
!> frontistr_amd/shim/hecmw_matvec_hip.f90 (hecmw_matvec).  INTEGRATION.md shows where a maintainer adds the three files.
module hecmw_hip_binding
  use iso_c_binding
  use hecmw_util
  implicit none
  private
  public :: fx_matrix_view, fx_comm_view, fx_solve_info
  public :: fx_solve, fx_matvec, fx_last_error, fx_solve_attempts, fx_solve_attempt_history
  ! device-side assembly / stress update driven by fistr1's own fstr_Newton (INTEGRATION.md section 5)
  public :: fx_mesh_view, fx_material_view, fx_nl_state_view
  public :: fx_upload, fx_solve_device_matrix, fx_nl_init_sections, fx_nl_stiffness_at, fx_nl_update_at, fx_nl_commit, fx_nl_get_state, &
            fx_nl_set_state, fx_assemble_c3d8_sections, fx_update_c3d8_linear, fx_update_c3d8_linear_prepare, fx_nl_snapshot
  public :: fxb_values_owner, fxb_values_addr
  public :: fxb_matrix_on_device, fxb_defer_bc, fxb_solve_device_matrix, FX_UP_PROFILE
  public :: fxb_context, fxb_views, fxb_ensure_transport, fxb_error_text, fxb_on_gpu_path, fx_get_stats

  type, bind(C) :: fx_matrix_view
    integer(c_int32_t) :: N, NP, NPL, NPU, NDOF
    type(c_ptr) :: indexL, itemL, indexU, itemU
    type(c_ptr) :: D, AL, AU, B, X
  end type fx_matrix_view

  type, bind(C) :: fx_comm_view
    integer(c_int32_t) :: my_rank, PETOT, nn_internal, n_node, n_neighbor_pe
    type(c_ptr) :: neighbor_pe, import_index, import_item, export_index, export_item
  end type fx_comm_view

  type, bind(C) :: fx_mesh_view
    integer(c_int32_t) :: n_node, n_elem
    type(c_ptr) :: coord, conn
  end type fx_mesh_view

  type, bind(C) :: fx_material_view
    real(c_double) :: E, nu
    integer(c_int32_t) :: plastic, harden, nlgeom, ntab
    real(c_double) :: plconst(3)
    type(c_ptr) :: tab
  end type fx_material_view

  type, bind(C) :: fx_nl_state_view
    type(c_ptr) :: stress, strain, stress_bak, strain_bak, plstrain, fstat, istat, unode, dunode, qforce
    integer(c_int32_t) :: latch
  end type fx_nl_state_view

  integer(c_int), parameter :: FX_UP_PROFILE = 1

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
    integer(c_int) function fx_upload(ctx, mat, comm, what) bind(C, name='fx_upload')
      import :: c_int, c_ptr, fx_matrix_view, fx_comm_view
      type(c_ptr), value :: ctx
      type(fx_matrix_view) :: mat
      type(fx_comm_view) :: comm
      integer(c_int), value :: what
    end function fx_upload
    integer(c_int) function fx_solve_device_matrix(ctx, mat, comm, n_bc, bc_node, bc_dof, bc_val, Iarray, Rarray, info, hist, hist_len) &
        bind(C, name='fx_solve_device_matrix')
      import :: c_int, c_ptr, c_int32_t, c_double, fx_matrix_view, fx_comm_view, fx_solve_info
      type(c_ptr), value :: ctx
      type(fx_matrix_view) :: mat
      type(fx_comm_view) :: comm
      integer(c_int32_t), value :: n_bc
      integer(c_int32_t) :: bc_node(*), bc_dof(*)
      real(c_double) :: bc_val(*)
      integer(c_int32_t) :: Iarray(100)
      real(c_double) :: Rarray(100)
      type(fx_solve_info) :: info
      real(c_double) :: hist(*)
      integer(c_int32_t), value :: hist_len
    end function fx_solve_device_matrix
    integer(c_int) function fx_assemble_c3d8_sections(ctx, mesh, n_mat, E, nu, elem_mat, elemopt, load, n_bc, bc_node, bc_dof, bc_val, ms) &
        bind(C, name='fx_assemble_c3d8_sections')
      import :: c_int, c_ptr, c_int32_t, c_double, c_float, fx_mesh_view
      type(c_ptr), value :: ctx
      type(fx_mesh_view) :: mesh
      integer(c_int32_t), value :: n_mat
      real(c_double) :: E(*), nu(*)
      integer(c_int32_t) :: elem_mat(*)
      integer(c_int), value :: elemopt
      type(c_ptr), value :: load                 ! NULL: B is not touched
      integer(c_int32_t), value :: n_bc
      type(c_ptr), value :: bc_node, bc_dof, bc_val
      real(c_float) :: ms
    end function fx_assemble_c3d8_sections
    integer(c_int) function fx_update_c3d8_linear(ctx, mesh, n_mat, E, nu, elem_mat, elemopt, disp, strain, stress, qforce, ms) &
        bind(C, name='fx_update_c3d8_linear')
      import :: c_ptr, c_int, c_int32_t, c_double, c_float, fx_mesh_view
      type(c_ptr), value :: ctx
      type(fx_mesh_view), intent(in) :: mesh
      integer(c_int32_t), value :: n_mat
      real(c_double), intent(in) :: E(*), nu(*)
      integer(c_int32_t), intent(in) :: elem_mat(*)
      integer(c_int), value :: elemopt
      real(c_double), intent(in) :: disp(*)
      type(c_ptr), intent(out) :: strain, stress          ! pinned host arrays of the library: (6, 8, n_elem)
      real(c_double), intent(inout) :: qforce(*)
      real(c_float), intent(out) :: ms
    end function fx_update_c3d8_linear
    integer(c_int) function fx_update_c3d8_linear_prepare(ctx, n_elem) bind(C, name='fx_update_c3d8_linear_prepare')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: n_elem
    end function fx_update_c3d8_linear_prepare
    integer(c_int) function fx_nl_init_sections(ctx, mesh, n_mat, mats, elem_mat) bind(C, name='fx_nl_init_sections')
      import :: c_int, c_ptr, c_int32_t, fx_mesh_view, fx_material_view
      type(c_ptr), value :: ctx
      type(fx_mesh_view) :: mesh
      integer(c_int32_t), value :: n_mat
      type(fx_material_view) :: mats(*)
      integer(c_int32_t) :: elem_mat(*)
    end function fx_nl_init_sections
    integer(c_int) function fx_nl_stiffness_at(ctx, unode, dunode, ms) bind(C, name='fx_nl_stiffness_at')
      import :: c_int, c_ptr, c_double, c_float
      type(c_ptr), value :: ctx
      real(c_double) :: unode(*), dunode(*)
      real(c_float) :: ms
    end function fx_nl_stiffness_at
    integer(c_int) function fx_nl_update_at(ctx, dunode, qforce, ms) bind(C, name='fx_nl_update_at')
      import :: c_int, c_ptr, c_double, c_float
      type(c_ptr), value :: ctx
      real(c_double) :: dunode(*), qforce(*)
      real(c_float) :: ms
    end function fx_nl_update_at
    integer(c_int) function fx_nl_commit(ctx) bind(C, name='fx_nl_commit')
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
    end function fx_nl_commit
    integer(c_int) function fx_nl_snapshot(ctx, load) bind(C, name='fx_nl_snapshot')
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int), value :: load
    end function fx_nl_snapshot
    integer(c_int) function fx_nl_get_state(ctx, s) bind(C, name='fx_nl_get_state')
      import :: c_int, c_ptr, fx_nl_state_view
      type(c_ptr), value :: ctx
      type(fx_nl_state_view) :: s
    end function fx_nl_get_state
    integer(c_int) function fx_nl_set_state(ctx, s) bind(C, name='fx_nl_set_state')
      import :: c_int, c_ptr, fx_nl_state_view
      type(c_ptr), value :: ctx
      type(fx_nl_state_view) :: s
    end function fx_nl_set_state
    integer(c_int) function fx_solve_attempts(ctx, cap, n_attempts, method, sigma_diag, n_hist) bind(C, name='fx_solve_attempts')
      import :: c_int, c_ptr, c_int32_t, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: cap
      integer(c_int32_t) :: n_attempts, method(*), n_hist(*)
      real(c_double) :: sigma_diag(*)
    end function fx_solve_attempts
    integer(c_int) function fx_solve_attempt_history(ctx, attempt, hist, cap) bind(C, name='fx_solve_attempt_history')
      import :: c_int, c_ptr, c_int32_t, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: attempt, cap
      real(c_double) :: hist(*)
    end function fx_solve_attempt_history
    integer(c_int) function fx_matvec(ctx, mat, comm, x, y, commtime) bind(C, name='fx_matvec')
      import :: c_int, c_ptr, c_double, fx_matrix_view, fx_comm_view
      type(c_ptr), value :: ctx
      type(fx_matrix_view) :: mat
      type(fx_comm_view) :: comm
      real(c_double) :: x(*), y(*)
      real(c_double) :: commtime
    end function fx_matvec
    integer(c_int) function fx_get_stats(ctx, out) bind(C, name='fx_get_stats')
      import :: c_ptr, c_int, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t) :: out(16)
    end function fx_get_stats
    function fx_last_error() bind(C, name='fx_last_error') result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function fx_last_error
    integer(c_int) function fx_comm_unique_id(id) bind(C, name='fx_comm_unique_id')
      import :: c_int, c_char
      character(kind=c_char) :: id(128)
    end function fx_comm_unique_id
    integer(c_int) function fx_comm_init(ctx, id, rank, nranks) bind(C, name='fx_comm_init')
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: ctx
      character(kind=c_char) :: id(128)
      integer(c_int), value :: rank, nranks
    end function fx_comm_init
    integer(c_int) function fx_comm_set_host_callbacks(ctx, rank, nranks, halo, allreduce, user) &
        bind(C, name='fx_comm_set_host_callbacks')
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: ctx
      integer(c_int), value :: rank, nranks
      type(c_funptr), value :: halo, allreduce
      type(c_ptr), value :: user
    end function fx_comm_set_host_callbacks
    function fxb_strlen(s) bind(C, name='strlen') result(n)
      import :: c_ptr, c_size_t
      type(c_ptr), value :: s
      integer(c_size_t) :: n
    end function fxb_strlen
  end interface

  type(c_ptr), save :: the_ctx = c_null_ptr
  ! The matrix of the next hecmw_solve was assembled ON the device (set by the fistr1-side binding of fstr_StiffMatrix): hecMAT%D / AL /
  ! AU on the host are stale, hecmw_mat_ass_bc only records its prescribed dofs (fxb_defer_bc), hecmw_solve applies them to the resident
  ! matrix and solves it (fxb_solve_device_matrix).
  logical, save :: fxb_matrix_on_device = .false.
  ! Whose matrix values the (single) device context of this rank holds: 0 nobody's, 1 the last hecmw_solve's, 2 the hecmw_matvec
  ! binding's (HECMW_GPU_MATVEC=resident re-uses them only while they are still its own: same owner, same hecMAT%D).
  integer, save :: fxb_values_owner = 0
  type(c_ptr), save :: fxb_values_addr = c_null_ptr
  integer(c_int32_t), allocatable, save :: dbc_node(:), dbc_dof(:)
  real(c_double), allocatable, save :: dbc_val(:)
  integer, save :: n_dbc = 0
  integer, save :: transport = 0              ! 0 none yet, 1 RCCL, 2 host callbacks through the reference's MPI layer
  ! what the callbacks need: the mesh whose tables they serve and the block size of the vector in flight
  type(hecmwST_local_mesh), pointer, save :: cb_mesh => null()
  integer(kind=kint), save :: cb_ndof = 3

contains

  !> ONE predicate for "this hecmw_solve call runs on the device": iterative solver, no MPC / contact matrix, a preconditioner and
  !> method the library has for this block size (include/fistr_hip.h).  hecmw_solve routes by it, and the fistr1-side binding of the
  !> element loops (fsd_eligible / fsd_eligible_linear) assembles on the device only when it holds -- a matrix assembled on the
  !> device can be solved nowhere else (the host D / AL / AU are never filled).
  logical function fxb_on_gpu_path(hecMESH, hecMAT)
    type(hecmwST_local_mesh), intent(in) :: hecMESH
    type(hecmwST_matrix), intent(in) :: hecMAT
    integer(kind=kint) :: precond
    precond = hecMAT%Iarray(3)
    fxb_on_gpu_path = hecMAT%Iarray(99) == 1 .and. hecMESH%mpc%n_mpc == 0 .and. hecMAT%cmat%n_val == 0
    if (hecMAT%NDOF == 3) then
      fxb_on_gpu_path = fxb_on_gpu_path .and. (precond == 1 .or. precond == 2 .or. precond == 3 .or. precond == 10)
    else   ! generic block sizes: METHOD 1-4 with SSOR / DIAG
      fxb_on_gpu_path = fxb_on_gpu_path .and. hecMAT%NDOF >= 1 .and. hecMAT%NDOF <= 6 .and. (precond >= 1 .and. precond <= 3) .and. &
                        (hecMAT%Iarray(2) >= 1 .and. hecMAT%Iarray(2) <= 4)
    endif
  end function fxb_on_gpu_path

  !> hecmw_mat_ass_bc's hook (three-line patch of hecmw_mat_ass.f90:292, INTEGRATION.md section 5): while the matrix lives on the
  !> device the prescribed dof is recorded for the solve instead of being eliminated from the (stale) host arrays.
  logical function fxb_defer_bc(inode, idof, rhs)
    integer(kind=kint), intent(in) :: inode, idof
    real(kind=kreal), intent(in) :: rhs
    integer(c_int32_t), allocatable :: ti(:)
    real(c_double), allocatable :: tr(:)
    integer :: cap
    fxb_defer_bc = fxb_matrix_on_device
    if (.not. fxb_matrix_on_device) return
    if (.not. allocated(dbc_node)) then
      allocate(dbc_node(1024), dbc_dof(1024), dbc_val(1024))
    else if (n_dbc >= size(dbc_node)) then
      cap = 2 * size(dbc_node)
      allocate(ti(cap)); ti(1:n_dbc) = dbc_node(1:n_dbc); call move_alloc(ti, dbc_node)
      allocate(ti(cap)); ti(1:n_dbc) = dbc_dof(1:n_dbc);  call move_alloc(ti, dbc_dof)
      allocate(tr(cap)); tr(1:n_dbc) = dbc_val(1:n_dbc);  call move_alloc(tr, dbc_val)
    endif
    n_dbc = n_dbc + 1
    dbc_node(n_dbc) = inode; dbc_dof(n_dbc) = idof; dbc_val(n_dbc) = rhs
  end function fxb_defer_bc

  !> hecmw_solve on the device-assembled matrix: right-hand side and X of hecMAT, the recorded prescribed dofs.
  function fxb_solve_device_matrix(ctx, mv, cv, hecMAT, info, hist, nhist) result(ierr)
    type(c_ptr), intent(in) :: ctx
    type(fx_matrix_view) :: mv
    type(fx_comm_view) :: cv
    type(hecmwST_matrix), target :: hecMAT
    type(fx_solve_info) :: info
    real(c_double) :: hist(*)
    integer(c_int32_t), intent(in) :: nhist
    integer(c_int) :: ierr
    integer(c_int32_t) :: none_i(1)
    real(c_double) :: none_r(1)
    if (n_dbc > 0) then
      ierr = fx_solve_device_matrix(ctx, mv, cv, int(n_dbc, c_int32_t), dbc_node, dbc_dof, dbc_val, hecMAT%Iarray, hecMAT%Rarray, &
                                    info, hist, nhist)
    else
      none_i = 0; none_r = 0.d0
      ierr = fx_solve_device_matrix(ctx, mv, cv, 0_c_int32_t, none_i, none_i, none_r, hecMAT%Iarray, hecMAT%Rarray, info, hist, nhist)
    endif
    n_dbc = 0
    fxb_matrix_on_device = .false.     ! consumed: the next fstr_StiffMatrix on the device raises it again; any other hecmw_mat_ass_bc / hecmw_solve sees host arrays
  end function fxb_solve_device_matrix

  !> The device context of this rank, created on first use (device = local rank: fx_create(-1)).
  function fxb_context(hecMESH) result(ctx)
    type(hecmwST_local_mesh), intent(in) :: hecMESH
    type(c_ptr) :: ctx
    integer(c_int) :: ierr
    if (.not. c_associated(the_ctx)) then
      ierr = fx_create(-1_c_int, the_ctx)
      if (ierr /= 0) then
        write(*,'(a,a)') '#### libfistr_hip: cannot create a device context: ', trim(fxb_error_text())
        call hecmw_abort(hecmw_comm_get_comm())
      endif
    endif
    ctx = the_ctx
  end function fxb_context

  function fxb_error_text() result(msg)
    character(len=512) :: msg
    character(kind=c_char), pointer :: p(:)
    type(c_ptr) :: cp
    integer :: i, n
    msg = ' '
    cp = fx_last_error()
    if (.not. c_associated(cp)) return
    n = min(int(fxb_strlen(cp)), len(msg))
    call c_f_pointer(cp, p, [n])
    do i = 1, n
      msg(i:i) = p(i)
    enddo
  end function fxb_error_text

  !> Borrowed views: c_loc of the members, nothing is copied (the caller owns every array, m_fstr.f90:807-857).
  subroutine fxb_views(hecMESH, hecMAT, mv, cv)
    type(hecmwST_local_mesh), intent(in), target :: hecMESH
    type(hecmwST_matrix), intent(in), target :: hecMAT
    type(fx_matrix_view), intent(out) :: mv
    type(fx_comm_view), intent(out) :: cv
    mv%N = hecMAT%N; mv%NP = hecMAT%NP; mv%NPL = hecMAT%NPL; mv%NPU = hecMAT%NPU; mv%NDOF = hecMAT%NDOF
    mv%indexL = c_loc(hecMAT%indexL(0)); mv%indexU = c_loc(hecMAT%indexU(0))
    mv%itemL = c_null_ptr; mv%itemU = c_null_ptr; mv%AL = c_null_ptr; mv%AU = c_null_ptr
    if (hecMAT%NPL > 0) then
      mv%itemL = c_loc(hecMAT%itemL(1)); mv%AL = c_loc(hecMAT%AL(1))
    endif
    if (hecMAT%NPU > 0) then
      mv%itemU = c_loc(hecMAT%itemU(1)); mv%AU = c_loc(hecMAT%AU(1))
    endif
    mv%D = c_loc(hecMAT%D(1)); mv%B = c_loc(hecMAT%B(1)); mv%X = c_loc(hecMAT%X(1))
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
  end subroutine fxb_views

  !> Establish the transport on the first call of a decomposed run (PETOT > 1, or a rank with neighbour tables); every
  !> call refreshes what the callbacks serve.  Collective over hecMESH%MPI_COMM the first time (the id broadcast).
  subroutine fxb_ensure_transport(hecMESH, ndof)
    use m_hecmw_comm_f
    type(hecmwST_local_mesh), intent(in), target :: hecMESH
    integer(kind=kint), intent(in) :: ndof
    character(kind=c_char) :: id(128)
    character(len=1) :: idc(128)
    character(len=16) :: env
    integer :: elen, estat, i
    integer(c_int) :: ierr
    type(c_ptr) :: ctx
    cb_mesh => hecMESH
    cb_ndof = ndof
    if (hecMESH%PETOT <= 1 .and. hecMESH%n_neighbor_pe <= 0) return
    if (transport /= 0) return
    ctx = fxb_context(hecMESH)
    call get_environment_variable('HECMW_GPU_TRANSPORT', env, elen, estat)
    if (estat == 0 .and. elen >= 3 .and. env(1:3) == 'mpi') then
      ierr = fx_comm_set_host_callbacks(ctx, int(hecMESH%my_rank, c_int), int(hecMESH%PETOT, c_int), &
                                        c_funloc(fxb_halo_cb), c_funloc(fxb_allreduce_cb), c_null_ptr)
      transport = 2
      if (hecMESH%my_rank == 0) write(*,'(a)') '### libfistr_hip: halo exchange and reductions through hecmw_update_m_R / hecmw_allreduce_R'
    else
      idc = ' '
      if (hecMESH%my_rank == 0) then
        ierr = fx_comm_unique_id(id)
        if (ierr /= 0) then
          write(*,'(a,a)') '#### libfistr_hip-E: cannot create an RCCL id (HECMW_GPU_TRANSPORT=mpi uses the MPI layer instead): ', &
            trim(fxb_error_text())
          call hecmw_abort(hecmw_comm_get_comm())
        endif
        do i = 1, 128
          idc(i) = id(i)
        enddo
      endif
      call hecmw_bcast_C(hecMESH, idc, 1, 128, 0)
      do i = 1, 128
        id(i) = idc(i)
      enddo
      ierr = fx_comm_init(ctx, id, int(hecMESH%my_rank, c_int), int(hecMESH%PETOT, c_int))
      transport = 1
      if (hecMESH%my_rank == 0 .and. ierr == 0) write(*,'(a,i0,a)') '### libfistr_hip: RCCL communicator over ', hecMESH%PETOT, ' rank(s)'
    endif
    if (ierr /= 0) then
      write(*,'(a,a)') '#### libfistr_hip-E: transport set-up failed: ', trim(fxb_error_text())
      call hecmw_abort(hecmw_comm_get_comm())
    endif
  end subroutine fxb_ensure_transport

  !> fx_halo_fn: `send` holds cb_ndof * n_export doubles in export_item order, `recv` receives cb_ndof * n_import doubles in
  !> import_item order.  The exchange itself is the reference's own hecmw_update_m_R on a nodal vector.
  subroutine fxb_halo_cb(send, recv, user) bind(C)
    use m_hecmw_comm_f
    real(c_double) :: send(*), recv(*)
    type(c_ptr), value :: user
    real(kind=kreal), allocatable :: val(:)
    integer(kind=kint) :: k, d, nd, np, ne, ni, node
    if (.not. associated(cb_mesh)) return
    nd = cb_ndof; np = cb_mesh%n_node
    ne = cb_mesh%export_index(cb_mesh%n_neighbor_pe)
    ni = cb_mesh%import_index(cb_mesh%n_neighbor_pe)
    allocate(val(nd * np))
    val = 0.d0
    do k = 1, ne
      node = cb_mesh%export_item(k)
      do d = 1, nd
        val(nd * (node - 1) + d) = send(nd * (k - 1) + d)
      enddo
    enddo
    call hecmw_update_m_R(cb_mesh, val, np, nd)
    do k = 1, ni
      node = cb_mesh%import_item(k)
      do d = 1, nd
        recv(nd * (k - 1) + d) = val(nd * (node - 1) + d)
      enddo
    enddo
    deallocate(val)
  end subroutine fxb_halo_cb

  !> fx_allreduce_fn: in-place SUM over the ranks (hecmw_allreduce_R, hecmw_comm_f.F90:346-379).
  subroutine fxb_allreduce_cb(v, n, user) bind(C)
    use m_hecmw_comm_f
    real(c_double) :: v(*)
    integer(c_int), value :: n
    type(c_ptr), value :: user
    real(kind=kreal), allocatable :: w(:)
    integer(kind=kint) :: i
    if (.not. associated(cb_mesh)) return
    allocate(w(n))
    do i = 1, n
      w(i) = v(i)
    enddo
    call hecmw_allreduce_R(cb_mesh, w, int(n, kint), hecmw_sum)
    do i = 1, n
      v(i) = w(i)
    enddo
    deallocate(w)
  end subroutine fxb_allreduce_cb

end module hecmw_hip_binding
