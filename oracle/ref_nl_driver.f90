!> TEST INFRASTRUCTURE ONLY (oracle).  Our own driver around the *reference's* nonlinear
!> (elastoplastic, updated/total Lagrange) C3D8 B-bar path:
!>   tangent          STF_C3D8Bbar      static_LIB_C3D8.f90:23   (called as fstr_StiffMatrix.f90:108-117)
!>   stress update    Update_C3D8Bbar   static_LIB_C3D8.f90:203  (called as fstr_Update.f90:155-163)
!>   return mapping   BackwardEuler     physics/Elastoplastic.f90:351 (inside Update_C3D8Bbar)
!>   state commit     updateEPState     physics/Elastoplastic.f90:563 + fstr_UpdateState fstr_Update.f90:296-345
!>   scatter / BC / solve  hecmw_mat_ass_elem, hecmw_mat_ass_bc, hecmw_solve (the reference's own)
!> The control flow of mode 2 (load factor ramp, Newton loop, residual, convergence test) is OUR
!> restatement of fstr_solve_NLGEOM.f90:100-121 / fstr_Newton (fstr_solve_NonLinear.f90:29-167) /
!> fstr_Update_NDForce (fstr_Residual.f90:23-71); every number inside it comes from reference routines.
!>
!> usage: ref_nl in.bin out.bin
!> in.bin : int32 magic(=1179209292) mode n_node n_elem n_bc nsub max_iter harden ntab nlgeom plastic
!>          real64 E nu pl1 pl2 pl3 converg ; real64 tab(2,ntab) (yield stress, plastic strain)
!>          int32 Iarray(100); real64 Rarray(100)
!>          real64 coord(3*n_node); int32 conn(8*n_elem)
!>          int32 bc_node(n_bc) bc_dof(n_bc); real64 bc_val(n_bc) (value at load factor 1)
!>          real64 cload(3*n_node) (nodal load at load factor 1)
!>   mode 1 only: real64 unode dunode (3*n_node each), stress_bak strain_bak stress strain (6,8,n_elem each),
!>          plstrain fstat1 (8,n_elem each); int32 istat(8,n_elem)
!> out.bin mode 1: real64 ke_before(24,24,n_elem) qf(24,n_elem) ke_after(24,24,n_elem)
!>                 real64 stress strain (6,8,n_elem), fstat1 (8,n_elem); int32 istat(8,n_elem)
!>         mode 2: int32 nlog; real64 log(7,nlog) = (sub, iter, cg_iter, res, xnrm, qnrm, dunrm)
!>                 real64 unode(3*n_node) qforce(3*n_node) stress strain (6,8,n_elem) plstrain fstat1 (8,n_elem)
!>                 int32 istat(8,n_elem)
program ref_nl
  use hecmw_util
  use hecmw_matrix_misc
  use hecmw_matrix_con
  use hecmw_matrix_ass
  use hecmw_solver
  use hecmw_solver_misc
  use m_table
  use mMaterial
  use mMechGauss
  use m_ElastoPlastic
  use m_static_LIB_C3D8
  implicit none
  type(hecmwST_local_mesh) :: hecMESH
  type(hecmwST_matrix)     :: hecMAT
  type(tMaterial), target  :: matl
  type(tMaterial), allocatable, target :: matls(:)   ! several sections: material 1 is the header's, 2.. follow the mesh data
  integer(kind=4) :: n_mat, harden2, ntab2, nlgeom2, plastic2, im
  integer(kind=4), allocatable :: elem_mat(:)
  real(kind=8) :: EE2, PP2, pl2(3)
  real(kind=8), allocatable :: tab2(:,:)
  type(tGaussStatus), allocatable :: gs(:,:)
  type(tTable) :: tbl
  character(len=1024) :: fin, fout
  integer(kind=4) :: magic, mode, n_node, n_elem, n_bc, nsub, max_iter, harden, ntab, nlgeom, plastic
  integer(kind=4) :: u, icel, j, i, k, sub, iter, nlog, lx
  real(kind=8) :: EE, PP, pl(3), converg, f1, f2, res, xnrm, qnrm, dunrm, rres, rxnrm
  real(kind=8), allocatable :: tab(:,:), coord(:), bc_val(:), cload(:), unode(:), dunode(:), qforce(:), GL(:)
  real(kind=8), allocatable :: a6(:,:,:), a1(:,:), kes(:,:,:), qfs(:,:), logv(:,:)
  integer(kind=4), allocatable :: conn(:), bc_node(:), bc_dof(:), ist(:,:)
  integer(kind=4) :: Iarr(100)
  real(kind=8) :: Rarr(100), fval(2,1)
  real(kind=8) :: stiff(24,24), ecoord(3,8), coords(3,3), uu(3,8), du(3,8), ut(3,8), qf(24)
  integer(kind=4) :: nodLOCAL(8)
  logical :: done

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, mode, n_node, n_elem, n_bc, nsub, max_iter, harden, ntab, nlgeom, plastic
  if (magic /= 1179209292) stop 'bad magic'
  read(u) EE, PP, pl, converg
  allocate(tab(2,max(ntab,1)))
  if (ntab > 0) read(u) tab(:,1:ntab)
  read(u) Iarr
  read(u) Rarr
  allocate(coord(3*n_node), conn(8*n_elem), bc_node(n_bc), bc_dof(n_bc), bc_val(n_bc), cload(3*n_node))
  allocate(unode(3*n_node), dunode(3*n_node), qforce(3*n_node), GL(3*n_node))
  read(u) coord
  read(u) conn
  read(u) bc_node
  read(u) bc_dof
  read(u) bc_val
  read(u) cload
  n_mat = 1
  if (mode >= 10) then    ! mode = 10 + mode: n_mat, then (EE, PP, pl, harden, ntab, nlgeom, plastic, tab) per further material, elem_mat
    mode = mode - 10
    read(u) n_mat
  endif
  allocate(matls(n_mat), elem_mat(n_elem))
  elem_mat = 1

  ! ---- material, as fstr_ctrl_get_ELASTICITY / fstr_ctrl_get_PLASTICITY leave it (fstr_ctrl_material.f90:60-106, :341-480)
  call make_material(matls(1), EE, PP, pl, harden, ntab, nlgeom, plastic, tab)
  do im = 2, n_mat
    read(u) EE2, PP2, pl2
    read(u) harden2, ntab2, nlgeom2, plastic2
    allocate(tab2(2,max(ntab2,1)))
    if (ntab2 > 0) read(u) tab2(:,1:ntab2)
    call make_material(matls(im), EE2, PP2, pl2, harden2, ntab2, nlgeom2, plastic2, tab2)
    deallocate(tab2)
  enddo
  if (n_mat > 1) read(u) elem_mat
  allocate(gs(8,n_elem))
  do icel = 1, n_elem
    do i = 1, 8
      gs(i,icel)%pMaterial => matls(elem_mat(icel))
      call fstr_init_gauss(gs(i,icel))
    enddo
  enddo
  coords = 0.d0
  allocate(a6(6,8,n_elem), a1(8,n_elem), ist(8,n_elem))

  if (mode == 1) then
    read(u) unode
    read(u) dunode
    read(u) a6; call put6(1)
    read(u) a6; call put6(2)
    read(u) a6; call put6(3)
    read(u) a6; call put6(4)
    read(u) a1
    do icel = 1, n_elem
      do i = 1, 8
        gs(i,icel)%plstrain = a1(i,icel)
      enddo
    enddo
    read(u) a1
    read(u) ist
    if (plastic == 1) then
      do icel = 1, n_elem
        do i = 1, 8
          gs(i,icel)%fstatus(1) = a1(i,icel)
          gs(i,icel)%istatus(1) = ist(i,icel)
        enddo
      enddo
    endif
    close(u)
    allocate(kes(24,24,n_elem), qfs(24,n_elem))
    open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
    do icel = 1, n_elem
      call gather(icel)
      ut = uu + du
      call STF_C3D8Bbar(361, 8, ecoord, gs(:,icel), kes(:,:,icel), 0, coords, 0.d0, 0.d0, ut)
    enddo
    write(u) kes
    do icel = 1, n_elem
      call gather(icel)
      call Update_C3D8Bbar(361, 8, ecoord, uu, du, 0, coords, qfs(:,icel), gs(:,icel), 1, 0.d0, 0.d0)
    enddo
    write(u) qfs
    do icel = 1, n_elem
      call gather(icel)
      ut = uu + du
      call STF_C3D8Bbar(361, 8, ecoord, gs(:,icel), kes(:,:,icel), 0, coords, 0.d0, 0.d0, ut)
    enddo
    write(u) kes
    call get6(1); write(u) a6
    call get6(2); write(u) a6
    call getst()
    write(u) a1
    write(u) ist
    close(u)
    stop
  endif
  close(u)

  ! ---- mode 2: load-step loop
  call hecmw_nullify_mesh(hecMESH)
  hecMESH%n_node = n_node; hecMESH%nn_internal = n_node; hecMESH%n_dof = 3
  hecMESH%n_elem = n_elem; hecMESH%n_elem_type = 1
  hecMESH%my_rank = 0; hecMESH%PETOT = 1; hecMESH%n_neighbor_pe = 0; hecMESH%mpc%n_mpc = 0
  allocate(hecMESH%elem_type_index(0:1), hecMESH%elem_type_item(1))
  hecMESH%elem_type_index(0) = 0; hecMESH%elem_type_index(1) = n_elem
  hecMESH%elem_type_item(1) = 361
  allocate(hecMESH%elem_node_index(0:n_elem), hecMESH%elem_node_item(8*n_elem))
  do i = 0, n_elem
    hecMESH%elem_node_index(i) = 8*i
  enddo
  hecMESH%elem_node_item = conn
  allocate(hecMESH%node(3*n_node))
  hecMESH%node = coord
  call hecmw_mat_init(hecMAT)
  hecMAT%NDOF = 3
  call hecmw_mat_con(hecMESH, hecMAT)
  allocate(hecMAT%D(9*hecMAT%NP), hecMAT%AL(9*hecMAT%NPL), hecMAT%AU(9*hecMAT%NPU))
  allocate(hecMAT%B(3*hecMAT%NP), hecMAT%X(3*hecMAT%NP))
  hecMAT%Iarray = Iarr
  hecMAT%Rarray = Rarr

  allocate(logv(7, nsub*max_iter))
  nlog = 0
  unode = 0.d0; qforce = 0.d0
  do sub = 1, nsub
    f1 = dble(sub-1)/dble(nsub)       ! table_nlsta without amplitude: linear ramp, fstr_solve_NLGEOM.f90:112-115
    f2 = dble(sub)/dble(nsub)
    dunode = 0.d0
    GL = cload*f2                      ! fstr_ass_load.f90:64-91
    hecMAT%B = GL - qforce             ! fstr_ass_load.f90:273
    done = .false.
    do iter = 1, max_iter
      call hecmw_mat_clear(hecMAT)     ! fstr_StiffMatrix.f90:38
      do icel = 1, n_elem
        call gather(icel)
        ut = uu + du
        call STF_C3D8Bbar(361, 8, ecoord, gs(:,icel), stiff, 0, coords, 0.d0, 0.d0, ut)
        call hecmw_mat_ass_elem(hecMAT, 8, nodLOCAL, stiff)
      enddo
      do k = 1, n_bc                   ! fstr_AddBC.f90:43-49,104
        if (iter == 1) then
          call hecmw_mat_ass_bc(hecMAT, bc_node(k), bc_dof(k), bc_val(k)*(f2-f1))
        else
          call hecmw_mat_ass_bc(hecMAT, bc_node(k), bc_dof(k), 0.d0)
        endif
      enddo
      if (sub == 1 .and. iter == 1) hecMAT%Iarray(98) = 1
      if (iter == 1) then
        hecMAT%Iarray(97) = 2
      else
        hecMAT%Iarray(97) = 1
      endif
      hecMAT%X = 0.d0
      call hecmw_solve(hecMESH, hecMAT)
      dunode = dunode + hecMAT%X(1:3*n_node)
      qforce = 0.d0                    ! fstr_UpdateNewton
      do icel = 1, n_elem
        call gather(icel)
        call Update_C3D8Bbar(361, 8, ecoord, uu, du, 0, coords, qf, gs(:,icel), iter, 0.d0, 0.d0)
        do j = 1, 8
          do i = 1, 3
            qforce(3*(nodLOCAL(j)-1)+i) = qforce(3*(nodLOCAL(j)-1)+i) + qf(3*(j-1)+i)
          enddo
        enddo
      enddo
      hecMAT%B(1:3*n_node) = GL - qforce      ! fstr_Update_NDForce
      do k = 1, n_bc
        hecMAT%B(3*(bc_node(k)-1)+bc_dof(k)) = 0.d0
      enddo
      call hecmw_InnerProduct_R(hecMESH, 3, hecMAT%B, hecMAT%B, res);  res = sqrt(res)
      call hecmw_InnerProduct_R(hecMESH, 3, hecMAT%X, hecMAT%X, xnrm); xnrm = sqrt(xnrm)
      call hecmw_InnerProduct_R(hecMESH, 3, qforce, qforce, qnrm);     qnrm = sqrt(qnrm)
      if (qnrm < 1.0d-8) qnrm = 1.0d0
      if (iter == 1) then
        dunrm = xnrm
      else
        call hecmw_InnerProduct_R(hecMESH, 3, dunode, dunode, dunrm);  dunrm = sqrt(dunrm)
      endif
      rres = res/qnrm
      rxnrm = xnrm/dunrm
      nlog = nlog + 1
      logv(:,nlog) = (/ dble(sub), dble(iter), dble(hecMAT%Iarray(1)), res, xnrm, qnrm, dunrm /)
      write(*,"(a,i4,a,i4,a,1pe11.4,a,1pe11.4)") " sub:", sub, " iter:", iter, ", residual:", rres, ", disp.corr.:", rxnrm
      if (hecMAT%Iarray(81) == 1) then      ! hecmw_mat_get_flag_converged
        if (rres < converg) done = .true.
        if (rxnrm < converg) done = .true.
      endif
      if (done) exit
    enddo
    unode = unode + dunode
    do icel = 1, n_elem                ! fstr_UpdateState
      do i = 1, 8
        if (isElastoplastic(gs(i,icel)%pMaterial%mtype)) call updateEPState(gs(i,icel))   ! fstr_Update.f90:323-326
        gs(i,icel)%strain_bak = gs(i,icel)%strain
        gs(i,icel)%stress_bak = gs(i,icel)%stress
      enddo
    enddo
  enddo

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) nlog
  write(u) logv(:,1:nlog)
  write(u) unode
  write(u) qforce
  call get6(1); write(u) a6
  call get6(2); write(u) a6
  do icel = 1, n_elem
    do i = 1, 8
      a1(i,icel) = gs(i,icel)%plstrain
    enddo
  enddo
  write(u) a1
  call getst()
  write(u) a1
  write(u) ist
  close(u)

contains

  subroutine make_material(m, E_, P_, pl_, harden_, ntab_, nlgeom_, plastic_, tab_)
    type(tMaterial), intent(inout) :: m
    real(kind=8), intent(in) :: E_, P_, pl_(3), tab_(:,:)
    integer(kind=4), intent(in) :: harden_, ntab_, nlgeom_, plastic_
    real(kind=8) :: fv(2,1)
    call initMaterial(m)
    m%mtype = ELASTIC
    m%nlgeom_flag = nlgeom_
    m%variables(M_YOUNGS) = E_
    m%variables(M_POISSON) = P_
    fv(1,1) = E_; fv(2,1) = P_
    call init_table(tbl, 0, 2, 1, fv)
    call dict_add_key(m%dict, MC_ISOELASTIC, tbl)
    call finalize_table(tbl)
    if (plastic_ == 1) then
      call setDigit(1, 1, m%mtype)
      call setDigit(2, 2, m%mtype)
      call setDigit(5, harden_, m%mtype)
      call setDigit(4, 0, m%mtype)
      m%variables(M_PLCONST1) = pl_(1)
      m%variables(M_PLCONST2) = pl_(2)
      m%variables(M_PLCONST3) = pl_(3)
      if (harden_ == 1) then
        call init_table(tbl, 1, 2, ntab_, tab_(:,1:ntab_))
        call dict_add_key(m%dict, MC_YIELD, tbl)
        call finalize_table(tbl)
      endif
    endif
  end subroutine make_material

  subroutine gather(ic)
    integer(kind=4), intent(in) :: ic
    integer(kind=4) :: jj, ii
    do jj = 1, 8
      nodLOCAL(jj) = conn(8*(ic-1)+jj)
      do ii = 1, 3
        ecoord(ii,jj) = coord(3*nodLOCAL(jj)+ii-3)
        uu(ii,jj) = unode(3*nodLOCAL(jj)+ii-3)
        du(ii,jj) = dunode(3*nodLOCAL(jj)+ii-3)
      enddo
    enddo
  end subroutine

  subroutine put6(which)
    integer(kind=4), intent(in) :: which
    integer(kind=4) :: ic, ii
    do ic = 1, n_elem
      do ii = 1, 8
        select case (which)
        case (1); gs(ii,ic)%stress_bak = a6(:,ii,ic)
        case (2); gs(ii,ic)%strain_bak = a6(:,ii,ic)
        case (3); gs(ii,ic)%stress = a6(:,ii,ic)
        case (4); gs(ii,ic)%strain = a6(:,ii,ic)
        end select
      enddo
    enddo
  end subroutine

  subroutine get6(which)
    integer(kind=4), intent(in) :: which
    integer(kind=4) :: ic, ii
    do ic = 1, n_elem
      do ii = 1, 8
        if (which == 1) then
          a6(:,ii,ic) = gs(ii,ic)%stress
        else
          a6(:,ii,ic) = gs(ii,ic)%strain
        endif
      enddo
    enddo
  end subroutine

  subroutine getst()
    integer(kind=4) :: ic, ii
    a1 = 0.d0; ist = 0
    do ic = 1, n_elem
      if (.not. isElastoplastic(gs(1,ic)%pMaterial%mtype)) cycle
      do ii = 1, 8
        a1(ii,ic) = gs(ii,ic)%fstatus(1)
        ist(ii,ic) = gs(ii,ic)%istatus(1)
      enddo
    enddo
  end subroutine

end program ref_nl
