!> TEST INFRASTRUCTURE ONLY (oracle).  Our own driver around the *reference's*
!> assembly path: CRS profile (hecmw_mat_con, hecmw_mat_con.f90:23), C3D8 element
!> stiffness (STF_C3D8IC static_LIB_3dIC.f90:21 | STF_C3D8Bbar static_LIB_C3D8.f90:23
!> | STF_C3 static_LIB_3d.f90:47), scatter (hecmw_mat_ass_elem hecmw_mat_ass.f90:31)
!> and Dirichlet elimination (hecmw_mat_ass_bc hecmw_mat_ass.f90:292), called the
!> way fstr_StiffMatrix.f90:58-207 / fstr_AddBC.f90 call them for a linear-elastic
!> TYPE=361 mesh with one isotropic material.
!>
!> usage: ref_fem in.bin out.bin
!> in.bin : int32 magic(=1179206989) elemopt(1=IC,2=Bbar,3=FI) n_node n_elem n_bc
!>          real64 E nu
!>          real64 coord(3*n_node); int32 conn(8*n_elem) (1-based)
!>          int32 bc_node(n_bc) bc_dof(n_bc); real64 bc_val(n_bc)
!>          real64 B0(3*n_node)
!>          elemopt > 10 (= 10 + formulation): several sections follow -- int32 n_mat ; real64 E(n_mat) nu(n_mat) ;
!>          int32 elem_mat(n_elem) (1-based material of each element, as hecMESH%section_ID -> fstrSOLID%materials)
!> out.bin: int32 N NP NPL NPU; int32 indexL(0:NP) indexU(0:NP) itemL itemU
!>          real64 D AL AU B ; real64 stiff_first_element(24,24) (column major)
!>          real64 t_assemble
program ref_fem
  use hecmw_util
  use hecmw_matrix_misc
  use hecmw_matrix_con
  use hecmw_matrix_ass
  use mMaterial
  use mMechGauss
  use m_static_LIB_3d
  use m_static_LIB_3dIC
  use m_static_LIB_C3D8
  implicit none
  type(hecmwST_local_mesh) :: hecMESH
  type(hecmwST_matrix)     :: hecMAT
  type(tMaterial), target  :: matl
  type(tMaterial), allocatable, target :: mats(:)
  integer(kind=4), allocatable :: elem_mat(:)
  real(kind=8), allocatable :: Es(:), nus(:)
  integer(kind=4) :: n_mat
  type(tGaussStatus)       :: gausses(8)
  character(len=1024) :: fin, fout
  integer(kind=4) :: magic, elemopt, n_node, n_elem, n_bc, u, icel, j, i, k
  real(kind=8) :: EE, PP, t0, t1
  real(kind=8), allocatable :: coord(:), bc_val(:), B0(:)
  integer(kind=4), allocatable :: conn(:), bc_node(:), bc_dof(:)
  real(kind=8) :: stiff(24,24), stiff1(24,24), ecoord(3,8), coords(3,3), uu(3,8)
  integer(kind=4) :: nodLOCAL(8)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, elemopt, n_node, n_elem, n_bc
  if (magic /= 1179206989) stop 'bad magic'
  read(u) EE, PP
  allocate(coord(3*n_node), conn(8*n_elem), bc_node(n_bc), bc_dof(n_bc), bc_val(n_bc), B0(3*n_node))
  read(u) coord
  read(u) conn
  read(u) bc_node
  read(u) bc_dof
  read(u) bc_val
  read(u) B0
  n_mat = 0
  if (elemopt > 10) then
    elemopt = elemopt - 10
    read(u) n_mat
    allocate(Es(n_mat), nus(n_mat), elem_mat(n_elem), mats(n_mat))
    read(u) Es
    read(u) nus
    read(u) elem_mat
    do i = 1, n_mat
      call initMaterial(mats(i))
      mats(i)%mtype = ELASTIC
      mats(i)%nlgeom_flag = INFINITE
      mats(i)%variables(M_YOUNGS) = Es(i)
      mats(i)%variables(M_POISSON) = nus(i)
    enddo
  endif
  close(u)

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
  hecMAT%X = 0.d0

  call initMaterial(matl)
  matl%mtype = ELASTIC
  matl%nlgeom_flag = INFINITE
  matl%variables(M_YOUNGS) = EE
  matl%variables(M_POISSON) = PP
  do i = 1, 8
    gausses(i)%pMaterial => matl
    call fstr_init_gauss(gausses(i))
  enddo

  t0 = hecmw_Wtime()
  call hecmw_mat_clear(hecMAT)
  uu = 0.d0
  coords = 0.d0
  do icel = 1, n_elem
    do j = 1, 8
      nodLOCAL(j) = conn(8*(icel-1)+j)
      do i = 1, 3
        ecoord(i,j) = coord(3*nodLOCAL(j)+i-3)
      enddo
    enddo
    if (n_mat > 0) then
      do i = 1, 8
        gausses(i)%pMaterial => mats(elem_mat(icel))
      enddo
    endif
    select case (elemopt)
    case (1)
      call STF_C3D8IC(361, 8, ecoord, gausses, stiff, 0, coords, 0.d0, 0.d0)
    case (2)
      call STF_C3D8Bbar(361, 8, ecoord, gausses, stiff, 0, coords, 0.d0, 0.d0, uu)
    case (3)
      call STF_C3(361, 8, ecoord, gausses, stiff, 0, coords, 0.d0, 0.d0, uu)
    case default
      stop 'bad elemopt'
    end select
    if (icel == 1) stiff1 = stiff
    call hecmw_mat_ass_elem(hecMAT, 8, nodLOCAL, stiff)
  enddo
  hecMAT%B = B0
  do k = 1, n_bc
    call hecmw_mat_ass_bc(hecMAT, bc_node(k), bc_dof(k), bc_val(k))
  enddo
  t1 = hecmw_Wtime()

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) hecMAT%N, hecMAT%NP, hecMAT%NPL, hecMAT%NPU
  write(u) hecMAT%indexL
  write(u) hecMAT%indexU
  write(u) hecMAT%itemL
  write(u) hecMAT%itemU
  write(u) hecMAT%D
  write(u) hecMAT%AL
  write(u) hecMAT%AU
  write(u) hecMAT%B
  write(u) stiff1
  write(u) t1 - t0
  close(u)
end program ref_fem
