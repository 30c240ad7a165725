!> fistr1-side half of the device-resident Newton iteration (INTEGRATION.md section 5): the element loops of fstr_StiffMatrix
!> (fistr1/src/analysis/static/fstr_StiffMatrix.f90:18-212) and fstr_UpdateNewton (fstr_Update.f90:25-293) and the history update
!> of fstr_UpdateState (:296-345) forwarded to libfistr_hip, so that a Newton iteration of fstr_Newton
!> (fstr_solve_NonLinear.f90:29-167) moves vectors of 3*NP doubles between host and device instead of the 6.5 GB matrix:
!>   fstr_StiffMatrix  -> fx_nl_stiffness_at (unode, dunode up; the tangent stays on the device)
!>   fstr_AddBC        -> unchanged; its hecmw_mat_ass_bc calls are recorded (hecmw_hip_binding: fxb_defer_bc)
!>   solve_LINEQ       -> hecmw_solve -> fx_solve_device_matrix (B, X and the prescribed dofs up, X down)
!>   fstr_UpdateNewton -> fx_nl_update_at (dunode up, QFORCE down)
!>   fstr_UpdateState  -> fx_nl_commit + the quadrature-point history down (once per sub-step: results, restart)
!> Taken only for what the device kernels cover -- static analysis with NLGEOM, every element TYPE=361 with the B-bar formulation,
!> isotropic ELASTIC or Mises-elastoplastic materials with isotropic hardening, no temperature / contact / MPC / spring / local
!> coordinate system / cutback; anything else runs the reference's own routines (kept, renamed, in the same binary).
!> HECMW_GPU_ASSEMBLY=0 keeps the reference's element loops on the host for every deck.
module fstr_device_hip
  use iso_c_binding
  use hecmw
  use m_fstr
  use mMechGauss
  use mMaterial
  use m_step
  use hecmw_hip_binding
  implicit none
  private
  public :: fsd_stiffness, fsd_update_newton, fsd_update_state, fsd_active, fsd_report, fsd_cutback

  logical, save :: decided = .false., eligible = .false., ready = .false.
  ! linear static decks: the stiffness loop (fx_assemble_c3d8_sections) and the stress update (fx_update_c3d8_linear)
  logical, save :: lin_decided = .false., lin_eligible = .false., lin_ready = .false.
  type(c_ptr), save :: the_ctx_saved = c_null_ptr
  integer(c_int), save :: lin_elemopt = 0
  real(c_double), allocatable, target, save :: lin_E(:), lin_nu(:)
  integer(c_int32_t), allocatable, target, save :: lin_emat(:)
  integer(c_int32_t), save :: n_elem = 0
  real(c_double), allocatable, target, save :: tabs(:,:,:)       ! (2, ntab_max, n_mat): the MC_YIELD tables handed to the library
  real(c_double), allocatable, target, save :: b6(:,:,:), b1(:,:), b6b(:,:,:)
  integer(c_int32_t), allocatable, target, save :: bi(:,:)

contains

  !> HECMW_GPU_REPORT=1: where each element loop ran and how long it took (wall clock), one line per call
  subroutine fsd_report(what, seconds)
    character(len=*), intent(in) :: what
    real(kind=kreal), intent(in) :: seconds
    character(len=8) :: env
    integer :: elen, estat
    logical, save :: asked = .false., on = .false.
    if (.not. asked) then
      asked = .true.
      call get_environment_variable('HECMW_GPU_REPORT', env, elen, estat)
      on = (estat == 0 .and. elen > 0 .and. env(1:1) == '1')
    endif
    if (on .and. hecmw_comm_get_rank() == 0) write(*,'(a,a,a,f10.3,a)') '### libfistr_hip: ', what, ': ', seconds, ' s'
  end subroutine fsd_report

  logical function fsd_active()
    fsd_active = ready
  end function fsd_active

  !> Is this run one the device kernels cover?  Decided once (the deck does not change during a run).
  logical function fsd_eligible(hecMESH, hecMAT, fstrSOLID)
    type(hecmwST_local_mesh), intent(in) :: hecMESH
    type(hecmwST_matrix), intent(in) :: hecMAT
    type(fstr_solid), intent(in) :: fstrSOLID
    character(len=8) :: env
    integer :: elen, estat, i, icel, cid
    if (decided) then
      fsd_eligible = eligible
      return
    endif
    decided = .true.
    eligible = .false.
    fsd_eligible = .false.
    call get_environment_variable('HECMW_GPU_ASSEMBLY', env, elen, estat)
    if (estat == 0 .and. elen > 0 .and. env(1:1) == '0') return
    call get_environment_variable('HECMW_GPU', env, elen, estat)
    if (estat == 0 .and. elen > 0 .and. env(1:1) == '0') return
    if (hecMAT%NDOF /= 3 .or. hecMESH%n_dof /= 3) return
    if (fstrPR%solution_type /= kstSTATIC .or. .not. fstrPR%nlgeom) return
    if (.not. fxb_on_gpu_path(hecMESH, hecMAT)) return                       ! the solve must run on the device too: same predicate as hecmw_solve (method, preconditioner, no MPC / contact)
    if (hecMESH%n_elem_type /= 1) return
    if (hecMESH%elem_type_item(1) /= 361) return
    if (hecMESH%mpc%n_mpc > 0) return
    if (fstrSOLID%TEMP_ngrp_tot > 0 .or. fstrSOLID%TEMP_irres > 0) return
    if (fstrSOLID%SPRING_ngrp_tot > 0) return
    if (associated(fstrSOLID%contacts)) then
      if (size(fstrSOLID%contacts) > 0) return
    endif
    if (fstrSOLID%n_fix_mpc > 0) return
    ! (automatic incrementation, `!AUTOINC_PARAM`: fstr_cutback_save / _load roll the device's copy of the history back with the
    !  host's, shim/fstr_Cutback_hip.f90 -> fsd_cutback.  A run continued from a restart file: the history read from the file is what
    !  fsd_init pushes to the device at the first fstr_StiffMatrix, after fstr_read_restart, fstr_solve_NLGEOM.f90:70-76.)
    do i = 1, hecMESH%section%n_sect
      if (fstrSOLID%sections(i)%elemopt361 /= kel361BBAR) return
      if (hecMESH%section%sect_orien_ID(i) > 0) return
    enddo
    do icel = 1, hecMESH%n_elem
      if (hecMESH%elem_node_index(icel) - hecMESH%elem_node_index(icel-1) /= 8) return
      cid = hecMESH%section%sect_mat_ID_item(hecMESH%section_ID(icel))
      if (.not. associated(fstrSOLID%elements(icel)%gausses(1)%pMaterial, fstrSOLID%materials(cid))) return
    enddo
    do i = 1, size(fstrSOLID%materials)
      if (.not. material_covered(fstrSOLID%materials(i))) return
    enddo
    eligible = .true.
    fsd_eligible = .true.
    if (hecMESH%my_rank == 0) write(*,'(a)') '### libfistr_hip: stiffness assembly and stress update on the device (TYPE=361 B-bar); '// &
      'HECMW_GPU_ASSEMBLY=0 keeps them on the host'
  end function fsd_eligible

  !> Linear static analysis (`!SOLUTION, TYPE=STATIC`, small strain) of TYPE=361 elements with isotropic ELASTIC materials: the
  !> element loop of fstr_StiffMatrix runs on the device -- STF_C3D8IC (the default of 361), STF_C3D8Bbar or STF_C3, whichever
  !> `!SECTION ... ELEMOPT361` selects, the same for every section -- and the matrix stays there for hecmw_solve; fstr_UpdateNewton
  !> (strains, stresses, QFORCE from the solution vector) remains the reference's routine, it never reads the matrix.
  logical function fsd_eligible_linear(hecMESH, hecMAT, fstrSOLID)
    type(hecmwST_local_mesh), intent(in) :: hecMESH
    type(hecmwST_matrix), intent(in) :: hecMAT
    type(fstr_solid), intent(in) :: fstrSOLID
    character(len=8) :: env
    integer :: elen, estat, i, icel, cid, opt
    if (lin_decided) then
      fsd_eligible_linear = lin_eligible
      return
    endif
    lin_decided = .true.
    lin_eligible = .false.
    fsd_eligible_linear = .false.
    call get_environment_variable('HECMW_GPU_ASSEMBLY', env, elen, estat)
    if (estat == 0 .and. elen > 0 .and. env(1:1) == '0') return
    call get_environment_variable('HECMW_GPU', env, elen, estat)
    if (estat == 0 .and. elen > 0 .and. env(1:1) == '0') return
    if (hecMAT%NDOF /= 3 .or. hecMESH%n_dof /= 3) return
    if (fstrPR%solution_type /= kstSTATIC .or. fstrPR%nlgeom) return
    if (.not. fxb_on_gpu_path(hecMESH, hecMAT)) return
    if (hecMESH%n_elem_type /= 1) return
    if (hecMESH%elem_type_item(1) /= 361) return
    if (hecMESH%mpc%n_mpc > 0) return
    if (fstrSOLID%TEMP_ngrp_tot > 0 .or. fstrSOLID%TEMP_irres > 0) return      ! thermal strains enter the element routine
    if (fstrSOLID%SPRING_ngrp_tot > 0) return
    if (associated(fstrSOLID%contacts)) then
      if (size(fstrSOLID%contacts) > 0) return
    endif
    if (fstrSOLID%n_fix_mpc > 0) return
    opt = -1
    do i = 1, hecMESH%section%n_sect
      if (opt == -1) opt = fstrSOLID%sections(i)%elemopt361
      if (fstrSOLID%sections(i)%elemopt361 /= opt) return
      if (hecMESH%section%sect_orien_ID(i) > 0) return
    enddo
    select case (opt)
      case (kel361IC);   lin_elemopt = 1
      case (kel361BBAR); lin_elemopt = 2
      case (kel361FI);   lin_elemopt = 3
      case default; return
    end select
    do icel = 1, hecMESH%n_elem
      if (hecMESH%elem_node_index(icel) - hecMESH%elem_node_index(icel-1) /= 8) return
      cid = hecMESH%section%sect_mat_ID_item(hecMESH%section_ID(icel))
      if (.not. associated(fstrSOLID%elements(icel)%gausses(1)%pMaterial, fstrSOLID%materials(cid))) return
    enddo
    do i = 1, size(fstrSOLID%materials)
      if (fstrSOLID%materials(i)%mtype == -1) cycle
      if (fstrSOLID%materials(i)%mtype /= ELASTIC) return
      if (fstrSOLID%materials(i)%nlgeom_flag /= INFINITE) return
      if (fetch_TableRow(MC_ISOELASTIC, fstrSOLID%materials(i)%dict) > 1) return     ! temperature-dependent constants
    enddo
    lin_eligible = .true.
    fsd_eligible_linear = .true.
    if (hecMESH%my_rank == 0) write(*,'(a)') '### libfistr_hip: stiffness assembly on the device (linear static, TYPE=361); '// &
      'HECMW_GPU_ASSEMBLY=0 keeps it on the host'
  end function fsd_eligible_linear

  subroutine fsd_init_linear(hecMESH, hecMAT, fstrSOLID)
    type(hecmwST_local_mesh), intent(in), target :: hecMESH
    type(hecmwST_matrix), intent(in), target :: hecMAT
    type(fstr_solid), intent(in) :: fstrSOLID
    type(c_ptr) :: ctx
    type(fx_matrix_view) :: mv
    type(fx_comm_view) :: cv
    integer(c_int) :: ierr
    integer :: i, icel, nmat
    ctx = fxb_context(hecMESH)
    call fxb_ensure_transport(hecMESH, 3)
    call fxb_views(hecMESH, hecMAT, mv, cv)
    mv%D = c_null_ptr; mv%AL = c_null_ptr; mv%AU = c_null_ptr; mv%B = c_null_ptr; mv%X = c_null_ptr   ! profile only
    ierr = fx_upload(ctx, mv, cv, FX_UP_PROFILE)
    if (ierr /= 0) call fsd_fail('profile upload')
    nmat = size(fstrSOLID%materials)
    if (allocated(lin_E)) deallocate(lin_E, lin_nu, lin_emat)
    allocate(lin_E(nmat), lin_nu(nmat), lin_emat(hecMESH%n_elem))
    lin_E = 1.d0; lin_nu = 0.d0
    do i = 1, nmat
      if (fstrSOLID%materials(i)%mtype == -1) cycle
      lin_E(i) = fstrSOLID%materials(i)%variables(M_YOUNGS)
      lin_nu(i) = fstrSOLID%materials(i)%variables(M_POISSON)
    enddo
    do icel = 1, hecMESH%n_elem
      lin_emat(icel) = hecMESH%section%sect_mat_ID_item(hecMESH%section_ID(icel))
    enddo
    lin_ready = .true.
  end subroutine fsd_init_linear

  logical function material_covered(m)
    type(tMaterial), intent(in) :: m
    material_covered = .false.
    if (m%mtype == -1) then            ! a slot of fstrSOLID%materials no section refers to (initMaterial never ran): ignored
      material_covered = .true.
      return
    endif
    if (m%nlgeom_flag /= INFINITE .and. m%nlgeom_flag /= TOTALLAG .and. m%nlgeom_flag /= UPDATELAG) return
    if (m%mtype == ELASTIC) then
      material_covered = .true.
    else if (isElastoplastic(m%mtype)) then
      if (getYieldFunction(m%mtype) /= 0) return         ! Mises
      if (isKinematicHarden(m%mtype)) return
      if (getHardenType(m%mtype) < 0 .or. getHardenType(m%mtype) > 3) return
      material_covered = .true.
    endif
  end function material_covered

  !> First use: profile, mesh, materials and the current quadrature-point history go to the device.
  subroutine fsd_init(hecMESH, hecMAT, fstrSOLID)
    type(hecmwST_local_mesh), intent(in), target :: hecMESH
    type(hecmwST_matrix), intent(in), target :: hecMAT
    type(fstr_solid), intent(inout), target :: fstrSOLID
    type(c_ptr) :: ctx
    type(fx_matrix_view) :: mv
    type(fx_comm_view) :: cv
    type(fx_mesh_view) :: mesh
    type(fx_material_view), allocatable :: mats(:)
    integer(c_int32_t), allocatable, target :: emat(:)
    type(DICT_DATA), pointer :: tbl
    logical :: ierr_l
    integer(c_int) :: ierr
    integer :: i, nmat, ntmax, nt, icel
    ctx = fxb_context(hecMESH)
    call fxb_ensure_transport(hecMESH, 3)
    call fxb_views(hecMESH, hecMAT, mv, cv)
    mv%D = c_null_ptr; mv%AL = c_null_ptr; mv%AU = c_null_ptr; mv%B = c_null_ptr; mv%X = c_null_ptr   ! profile only
    ierr = fx_upload(ctx, mv, cv, FX_UP_PROFILE)
    if (ierr /= 0) call fsd_fail('profile upload')
    n_elem = hecMESH%n_elem
    mesh%n_node = hecMESH%n_node; mesh%n_elem = n_elem
    mesh%coord = c_loc(hecMESH%node(1)); mesh%conn = c_loc(hecMESH%elem_node_item(1))
    nmat = size(fstrSOLID%materials)
    allocate(mats(nmat), emat(n_elem))
    ntmax = 1
    do i = 1, nmat
      if (fstrSOLID%materials(i)%mtype == -1) cycle
      if (isElastoplastic(fstrSOLID%materials(i)%mtype)) then
        if (getHardenType(fstrSOLID%materials(i)%mtype) == 1) ntmax = max(ntmax, fetch_TableRow(MC_YIELD, fstrSOLID%materials(i)%dict))
      endif
    enddo
    if (allocated(tabs)) deallocate(tabs)
    allocate(tabs(2, ntmax, nmat))
    tabs = 0.d0
    do i = 1, nmat
      mats(i)%E = 1.d0; mats(i)%nu = 0.d0; mats(i)%plastic = 0; mats(i)%harden = 0; mats(i)%nlgeom = 0; mats(i)%ntab = 0
      mats(i)%plconst = 0.d0; mats(i)%tab = c_null_ptr
      if (fstrSOLID%materials(i)%mtype == -1) cycle
      mats(i)%E = fstrSOLID%materials(i)%variables(M_YOUNGS)
      mats(i)%nu = fstrSOLID%materials(i)%variables(M_POISSON)
      mats(i)%nlgeom = fstrSOLID%materials(i)%nlgeom_flag
      if (isElastoplastic(fstrSOLID%materials(i)%mtype)) then
        mats(i)%plastic = 1
        mats(i)%harden = getHardenType(fstrSOLID%materials(i)%mtype)
        mats(i)%plconst(1) = fstrSOLID%materials(i)%variables(M_PLCONST1)
        mats(i)%plconst(2) = fstrSOLID%materials(i)%variables(M_PLCONST2)
        mats(i)%plconst(3) = fstrSOLID%materials(i)%variables(M_PLCONST3)
        if (mats(i)%harden == 1) then      ! MULTILINEAR: the MC_YIELD table as the reference holds it (tbval(1:2, 1:rows))
          call fetch_Table(MC_YIELD, fstrSOLID%materials(i)%dict, tbl, ierr_l)
          if (ierr_l) call fsd_fail('MC_YIELD table of a MULTILINEAR material')
          nt = tbl%tbrow
          tabs(1:2, 1:nt, i) = tbl%tbval(1:2, 1:nt)
          mats(i)%ntab = nt
          mats(i)%tab = c_loc(tabs(1, 1, i))
        endif
      endif
    enddo
    do icel = 1, n_elem
      emat(icel) = hecMESH%section%sect_mat_ID_item(hecMESH%section_ID(icel))
    enddo
    ierr = fx_nl_init_sections(ctx, mesh, int(nmat, c_int32_t), mats, emat)
    if (ierr /= 0) call fsd_fail('fx_nl_init_sections')
    deallocate(mats, emat)
    if (allocated(b6)) deallocate(b6, b6b, b1, bi)
    allocate(b6(6, 8, n_elem), b6b(6, 8, n_elem), b1(8, n_elem), bi(8, n_elem))
    call fsd_push_state(ctx, fstrSOLID)
    ready = .true.
    the_ctx_saved = ctx
    if (fstr_cutback_is_on(fstrSOLID)) then    ! the state fstr_solve_NLGEOM.f90:84 saved on the host before the device had any
      ierr = fx_nl_snapshot(ctx, 0_c_int)
      if (ierr /= 0) call fsd_fail('fx_nl_snapshot')
    endif
  end subroutine fsd_init

  type(c_ptr) function the_device_ctx()
    the_device_ctx = the_ctx_saved
  end function the_device_ctx

  logical function fstr_cutback_is_on(fstrSOLID)       ! is_cutback_active of fstr_Cutback.f90:32-34
    type(fstr_solid), intent(in) :: fstrSOLID
    integer :: i
    fstr_cutback_is_on = .false.
    do i = 1, fstrSOLID%nstep_tot
      if (fstrSOLID%step_ctrl(i)%inc_type == stepAutoInc) fstr_cutback_is_on = .true.
    enddo
  end function fstr_cutback_is_on

  subroutine fsd_fail(what)
    character(len=*), intent(in) :: what
    write(*,'(a,a,a,a)') '#### libfistr_hip-E: device assembly binding: ', what, ': ', trim(fxb_error_text())
    call hecmw_abort(hecmw_comm_get_comm())
  end subroutine fsd_fail

  !> host quadrature-point history -> device (initial state, or a state read from a restart file)
  subroutine fsd_push_state(ctx, fstrSOLID)
    type(c_ptr), intent(in) :: ctx
    type(fstr_solid), intent(inout), target :: fstrSOLID
    type(fx_nl_state_view) :: sv
    integer(c_int) :: ierr
    integer :: icel, g
    real(c_double), allocatable, target :: s6(:,:,:), sb6(:,:,:), e6b(:,:,:), pl(:,:), fs(:,:)
    allocate(s6(6, 8, n_elem), sb6(6, 8, n_elem), e6b(6, 8, n_elem), pl(8, n_elem), fs(8, n_elem))
    bi = 0; fs = 0.d0
    do icel = 1, n_elem
      do g = 1, 8
        b6(:, g, icel)  = fstrSOLID%elements(icel)%gausses(g)%strain
        s6(:, g, icel)  = fstrSOLID%elements(icel)%gausses(g)%stress
        e6b(:, g, icel) = fstrSOLID%elements(icel)%gausses(g)%strain_bak
        sb6(:, g, icel) = fstrSOLID%elements(icel)%gausses(g)%stress_bak
        pl(g, icel)     = fstrSOLID%elements(icel)%gausses(g)%plstrain
        if (associated(fstrSOLID%elements(icel)%gausses(g)%istatus)) bi(g, icel) = fstrSOLID%elements(icel)%gausses(g)%istatus(1)
        if (associated(fstrSOLID%elements(icel)%gausses(g)%fstatus)) fs(g, icel) = fstrSOLID%elements(icel)%gausses(g)%fstatus(1)
      enddo
    enddo
    sv%stress = c_loc(s6(1,1,1)); sv%strain = c_loc(b6(1,1,1)); sv%stress_bak = c_loc(sb6(1,1,1)); sv%strain_bak = c_loc(e6b(1,1,1))
    sv%plstrain = c_loc(pl(1,1)); sv%fstat = c_loc(fs(1,1)); sv%istat = c_loc(bi(1,1))
    sv%unode = c_loc(fstrSOLID%unode(1)); sv%dunode = c_loc(fstrSOLID%dunode(1)); sv%qforce = c_loc(fstrSOLID%QFORCE(1))
    sv%latch = -1
    ierr = fx_nl_set_state(ctx, sv)
    if (ierr /= 0) call fsd_fail('fx_nl_set_state')
    deallocate(s6, sb6, e6b, pl, fs)
  end subroutine fsd_push_state

  !> fstr_StiffMatrix on the device; .false. = not covered, the caller runs the reference's routine.
  logical function fsd_stiffness(hecMESH, hecMAT, fstrSOLID)
    type(hecmwST_local_mesh), intent(in), target :: hecMESH
    type(hecmwST_matrix), intent(inout), target :: hecMAT
    type(fstr_solid), intent(inout), target :: fstrSOLID
    integer(c_int) :: ierr
    real(c_float) :: ms
    type(fx_mesh_view) :: mesh
    character(len=8) :: env
    integer :: elen, estat
    fsd_stiffness = .false.
    fxb_matrix_on_device = .false.
    if (.not. fsd_eligible(hecMESH, hecMAT, fstrSOLID)) then
      if (.not. fsd_eligible_linear(hecMESH, hecMAT, fstrSOLID)) return
      if (.not. lin_ready) call fsd_init_linear(hecMESH, hecMAT, fstrSOLID)
      mesh%n_node = hecMESH%n_node; mesh%n_elem = hecMESH%n_elem
      mesh%coord = c_loc(hecMESH%node(1)); mesh%conn = c_loc(hecMESH%elem_node_item(1))
      ierr = fx_assemble_c3d8_sections(fxb_context(hecMESH), mesh, int(size(lin_E), c_int32_t), lin_E, lin_nu, lin_emat, lin_elemopt, &
                                       c_null_ptr, 0_c_int32_t, c_null_ptr, c_null_ptr, c_null_ptr, ms)
      if (ierr /= 0) call fsd_fail('fx_assemble_c3d8_sections')
      call get_environment_variable('HECMW_GPU_UPDATE', env, elen, estat)
      if (.not. (estat == 0 .and. elen > 0 .and. env(1:1) == '0')) &    ! the stress update follows the solve: pin its staging meanwhile
        ierr = fx_update_c3d8_linear_prepare(fxb_context(hecMESH), int(hecMESH%n_elem, c_int32_t))
      fxb_matrix_on_device = .true.
      fsd_stiffness = .true.
      return
    endif
    if (.not. ready) call fsd_init(hecMESH, hecMAT, fstrSOLID)
    ierr = fx_nl_stiffness_at(fxb_context(hecMESH), fstrSOLID%unode, fstrSOLID%dunode, ms)
    if (ierr /= 0) call fsd_fail('fx_nl_stiffness_at')
    fxb_matrix_on_device = .true.       ! from here to the solve: hecmw_mat_ass_bc records, hecmw_solve uses the resident matrix
    fsd_stiffness = .true.
  end function fsd_stiffness

  !> fstr_UpdateNewton on the device (QFORCE comes back, then the caller's halo update as fstr_Update.f90:284).
  logical function fsd_update_newton(hecMESH, fstrSOLID)
    type(hecmwST_local_mesh), intent(in) :: hecMESH
    type(fstr_solid), intent(inout), target :: fstrSOLID
    integer(c_int) :: ierr
    real(c_float) :: ms
    fsd_update_newton = .false.
    if (.not. ready) then
      if (lin_ready) fsd_update_newton = fsd_update_newton_linear(hecMESH, fstrSOLID)
      return
    endif
    ierr = fx_nl_update_at(fxb_context(hecMESH), fstrSOLID%dunode, fstrSOLID%QFORCE, ms)
    if (ierr /= 0) call fsd_fail('fx_nl_update_at')
    call hecmw_update_3_R(hecMESH, fstrSOLID%QFORCE, hecMESH%n_node)
    fsd_update_newton = .true.
  end function fsd_update_newton

  !> fstr_cutback_save (load = 0) / fstr_cutback_load (load = 1) for the device's copy of the quadrature-point history.  Before the
  !> first fstr_StiffMatrix (fstr_solve_NLGEOM.f90:84 saves the initial state) there is nothing on the device yet: fsd_init will
  !> push the host's state, which is that very state, and takes the first snapshot itself.
  subroutine fsd_cutback(load)
    integer, intent(in) :: load
    integer(c_int) :: ierr
    if (.not. ready) return
    ierr = fx_nl_snapshot(the_device_ctx(), int(load, c_int))
    if (ierr /= 0) call fsd_fail('fx_nl_snapshot')
  end subroutine fsd_cutback

  !> fstr_UpdateNewton of a linear static deck (fsd_eligible_linear: TYPE=361, isotropic ELASTIC, the formulation of `ELEMOPT361`):
  !> UpdateST_C3D8IC / Update_C3D8Bbar / UPDATE_C3 for every element on the device from the total displacement unode + dunode
  !> (fstr_Update.f90:165 for IC; static_LIB_3d.f90:556 / static_LIB_C3D8.f90:258 for the others); strain and stress come back through
  !> the library's pinned staging into fstrSOLID%elements(:)%gausses(:), QFORCE into fstrSOLID%QFORCE, then the caller-side halo update
  !> of fstr_Update.f90:284.  HECMW_GPU_UPDATE=0 keeps the reference's element loop.
  logical function fsd_update_newton_linear(hecMESH, fstrSOLID)
    type(hecmwST_local_mesh), intent(in), target :: hecMESH
    type(fstr_solid), intent(inout), target :: fstrSOLID
    integer(c_int) :: ierr
    real(c_float) :: ms
    type(fx_mesh_view) :: mesh
    type(c_ptr) :: ps, pt
    real(c_double), pointer :: s6(:,:,:), t6(:,:,:)
    real(c_double), allocatable :: tot(:)
    character(len=8) :: env
    integer :: elen, estat, icel, g
    real(kind=kreal) :: t0
    fsd_update_newton_linear = .false.
    call get_environment_variable('HECMW_GPU_UPDATE', env, elen, estat)
    if (estat == 0 .and. elen > 0 .and. env(1:1) == '0') return
    allocate(tot(3*hecMESH%n_node))
    tot(:) = fstrSOLID%unode(1:3*hecMESH%n_node) + fstrSOLID%dunode(1:3*hecMESH%n_node)
    mesh%n_node = hecMESH%n_node; mesh%n_elem = hecMESH%n_elem
    mesh%coord = c_loc(hecMESH%node(1)); mesh%conn = c_loc(hecMESH%elem_node_item(1))
    t0 = hecmw_Wtime()
    ierr = fx_update_c3d8_linear(fxb_context(hecMESH), mesh, int(size(lin_E), c_int32_t), lin_E, lin_nu, lin_emat, lin_elemopt, tot, &
                                 ps, pt, fstrSOLID%QFORCE, ms)
    if (ierr /= 0) call fsd_fail('fx_update_c3d8_linear')
    deallocate(tot)
    call fsd_report('  of which the library call (uploads, kernel, strain / stress / QFORCE back)', hecmw_Wtime() - t0)
    call fsd_report('  of which the element kernel alone', real(ms, kreal) * 1.d-3)
    call c_f_pointer(ps, s6, [6, 8, hecMESH%n_elem])
    call c_f_pointer(pt, t6, [6, 8, hecMESH%n_elem])
    !$omp parallel do default(shared) private(icel, g)
    do icel = 1, hecMESH%n_elem
      do g = 1, 8
        fstrSOLID%elements(icel)%gausses(g)%strain(1:6) = s6(1:6, g, icel)
        fstrSOLID%elements(icel)%gausses(g)%stress(1:6) = t6(1:6, g, icel)
      enddo
    enddo
    !$omp end parallel do
    call hecmw_update_3_R(hecMESH, fstrSOLID%QFORCE, hecMESH%n_node)
    fsd_update_newton_linear = .true.
  end function fsd_update_newton_linear

  !> fstr_UpdateState on the device, then the history comes back to fstrSOLID%elements (results, restart files and whatever else
  !> of fistr1 reads it) -- once per sub-step, not per Newton iteration.
  logical function fsd_update_state(hecMESH, fstrSOLID)
    type(hecmwST_local_mesh), intent(in) :: hecMESH
    type(fstr_solid), intent(inout), target :: fstrSOLID
    type(fx_nl_state_view) :: sv
    type(c_ptr) :: ctx
    integer(c_int) :: ierr
    integer :: icel, g
    real(c_double), allocatable, target :: s6(:,:,:), pl(:,:), fs(:,:)
    fsd_update_state = .false.
    if (.not. ready) return
    ctx = fxb_context(hecMESH)
    ierr = fx_nl_commit(ctx)      ! unode += dunode is the host's (fstr_Newton :156-158); the device does the same on its copy
    if (ierr /= 0) call fsd_fail('fx_nl_commit')
    allocate(s6(6, 8, n_elem), pl(8, n_elem), fs(8, n_elem))
    sv%stress = c_loc(s6(1,1,1)); sv%strain = c_loc(b6(1,1,1)); sv%stress_bak = c_null_ptr; sv%strain_bak = c_null_ptr
    sv%plstrain = c_loc(pl(1,1)); sv%fstat = c_loc(fs(1,1)); sv%istat = c_loc(bi(1,1))
    sv%unode = c_null_ptr; sv%dunode = c_null_ptr; sv%qforce = c_null_ptr
    sv%latch = -1
    ierr = fx_nl_get_state(ctx, sv)
    if (ierr /= 0) call fsd_fail('fx_nl_get_state')
    do icel = 1, n_elem
      do g = 1, 8
        fstrSOLID%elements(icel)%gausses(g)%strain = b6(:, g, icel)
        fstrSOLID%elements(icel)%gausses(g)%stress = s6(:, g, icel)
        fstrSOLID%elements(icel)%gausses(g)%strain_bak = b6(:, g, icel)     ! fstr_UpdateState :338-339
        fstrSOLID%elements(icel)%gausses(g)%stress_bak = s6(:, g, icel)
        fstrSOLID%elements(icel)%gausses(g)%plstrain = pl(g, icel)
        if (associated(fstrSOLID%elements(icel)%gausses(g)%istatus)) fstrSOLID%elements(icel)%gausses(g)%istatus(1) = bi(g, icel)
        if (associated(fstrSOLID%elements(icel)%gausses(g)%fstatus)) fstrSOLID%elements(icel)%gausses(g)%fstatus(1) = fs(g, icel)
      enddo
    enddo
    deallocate(s6, pl, fs)
    fsd_update_state = .true.
  end function fsd_update_state

end module fstr_device_hip
