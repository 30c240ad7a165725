!> TEST INFRASTRUCTURE ONLY (oracle).  Our own driver around the *reference's* stress-update routines of a linear static analysis,
!> called the way fstr_UpdateNewton (fistr1/src/analysis/static/fstr_Update.f90:73-264) calls them for a TYPE=361 mesh of
!> isotropic ELASTIC materials (nlgeom_flag INFINITE, no temperature, no material coordinate system):
!>   elemopt 1  UpdateST_C3D8IC  static_LIB_3dIC.f90:220-455  (total_disp = unode + dunode, :165)
!>   elemopt 2  Update_C3D8Bbar  static_LIB_C3D8.f90:203-547  (u = unode, du = dunode)
!>   elemopt 3  UPDATE_C3        static_LIB_3d.f90:516-837
!> and the QFORCE accumulation of :258-264.
!>
!> usage: ref_update in.bin out.bin
!> in.bin : int32 magic(=1179210064) elemopt n_node n_elem n_mat
!>          real64 E(n_mat) nu(n_mat) ; int32 elem_mat(n_elem) (1-based)
!>          real64 coord(3*n_node) ; int32 conn(8*n_elem) (1-based) ; real64 unode(3*n_node) dunode(3*n_node)
!> out.bin: real64 strain(6,8,n_elem) stress(6,8,n_elem) qforce(3*n_node)
program ref_update
  use hecmw_util
  use mMaterial
  use mMechGauss
  use m_static_LIB_3d
  use m_static_LIB_3dIC
  use m_static_LIB_C3D8
  implicit none
  type(tMaterial), allocatable, target :: mats(:)
  type(tGaussStatus) :: gausses(8)
  character(len=1024) :: fin, fout
  integer(kind=4) :: magic, elemopt, n_node, n_elem, n_mat, u, icel, j, i
  real(kind=8), allocatable :: Es(:), nus(:), coord(:), unode(:), dunode(:), strain(:,:,:), stress(:,:,:), qforce(:)
  integer(kind=4), allocatable :: conn(:), elem_mat(:)
  real(kind=8) :: ecoord(3,8), coords(3,3), total_disp(3,8), du(3,8), qf(24)
  integer(kind=4) :: nodLOCAL(8)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, elemopt, n_node, n_elem, n_mat
  if (magic /= 1179210064) stop 'bad magic'
  allocate(Es(n_mat), nus(n_mat), elem_mat(n_elem), mats(n_mat))
  allocate(coord(3*n_node), conn(8*n_elem), unode(3*n_node), dunode(3*n_node))
  read(u) Es
  read(u) nus
  read(u) elem_mat
  read(u) coord
  read(u) conn
  read(u) unode
  read(u) dunode
  close(u)
  do i = 1, n_mat
    call initMaterial(mats(i))
    mats(i)%mtype = ELASTIC
    mats(i)%nlgeom_flag = INFINITE
    mats(i)%variables(M_YOUNGS) = Es(i)
    mats(i)%variables(M_POISSON) = nus(i)
  enddo
  do i = 1, 8
    gausses(i)%pMaterial => mats(1)
    call fstr_init_gauss(gausses(i))
  enddo
  allocate(strain(6,8,n_elem), stress(6,8,n_elem), qforce(3*n_node))
  qforce = 0.d0
  coords = 0.d0
  do icel = 1, n_elem
    do j = 1, 8
      nodLOCAL(j) = conn(8*(icel-1)+j)
      do i = 1, 3
        ecoord(i,j) = coord(3*nodLOCAL(j)+i-3)
        total_disp(i,j) = unode(3*nodLOCAL(j)+i-3)
        du(i,j) = dunode(3*nodLOCAL(j)+i-3)
      enddo
    enddo
    do i = 1, 8
      gausses(i)%pMaterial => mats(elem_mat(icel))
    enddo
    select case (elemopt)
    case (1)
      total_disp(1:3, 1:8) = total_disp(1:3, 1:8) + du(1:3, 1:8)          ! fstr_Update.f90:165
      call UpdateST_C3D8IC(361, 8, ecoord(1, 1:8), ecoord(2, 1:8), ecoord(3, 1:8), total_disp(1:3, 1:8), gausses, 0, coords, qf=qf)
    case (2)
      call Update_C3D8Bbar(361, 8, ecoord, total_disp, du, 0, coords, qf, gausses, 1, 0.d0, 0.d0)
    case (3)
      call UPDATE_C3(361, 8, ecoord, total_disp, du, 0, coords, qf, gausses, 1, 0.d0, 0.d0)
    case default
      stop 'bad elemopt'
    end select
    do i = 1, 8
      strain(1:6, i, icel) = gausses(i)%strain(1:6)
      stress(1:6, i, icel) = gausses(i)%stress(1:6)
    enddo
    do j = 1, 8
      do i = 1, 3
        qforce(3*(nodLOCAL(j)-1)+i) = qforce(3*(nodLOCAL(j)-1)+i) + qf(3*(j-1)+i)
      enddo
    enddo
  enddo
  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) strain
  write(u) stress
  write(u) qforce
  close(u)
end program ref_update
