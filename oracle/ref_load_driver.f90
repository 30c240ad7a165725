!> TEST INFRASTRUCTURE ONLY (oracle).  Our own driver around the *reference's* distributed-load routine DL_C3
!> (fistr1/src/lib/static_LIB_3d.f90:210-377), called element by element as fstr_ass_load.f90:140-256 does, to produce
!> the nodal load vector of the reference's example decks exB..exE (pressure / body force / gravity / centrifugal).
!> Load assembly itself is outside the hot path (SURVEY §8a); the vector is fixture INPUT for the known-answer tests
!> of assembly + solve.
!> usage: ref_load in.bin out.bin
!> in.bin : int32 magic(=1179208772) n_node n_elem n_load ; real64 coord(3*n_node) ; int32 conn(8*n_elem)
!>          per load: int32 ltype n_el ; real64 params(0:6) rho ; int32 elems(n_el) (1-based)
!>          ltype = -1: thermal load of the IC element, TLOAD_C3D8IC (static_LIB_3dIC.f90:460-625, called as
!>          fstr_ass_load.f90:379-390): params = (T, T0, alpha, E, nu, REFTEMP, -), uniform temperatures
!> out.bin: real64 GL(3*n_node)
program ref_load
  use hecmw_util
  use mMaterial
  use mMechGauss
  use m_static_LIB_3d
  use m_static_LIB_3dIC
  use m_fstr, only: REF_TEMP
  implicit none
  real(kind=8), target :: reftemp_store
  type(tMaterial), target :: matl
  type(tGaussStatus) :: gausses(8)
  real(kind=8) :: tt(8), t0(8), coords(3,3)
  character(len=1024) :: fin, fout
  integer(kind=4) :: magic, n_node, n_elem, n_load, u, il, ltype, n_el, k, icel, j, i, nsize
  real(kind=8), allocatable :: coord(:), GL(:)
  integer(kind=4), allocatable :: conn(:), elems(:)
  real(kind=8) :: params(0:6), rho, xx(8), yy(8), zz(8), vect(24)
  integer(kind=4) :: nod(8)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, n_node, n_elem, n_load
  if (magic /= 1179208772) stop 'bad magic'
  allocate(coord(3*n_node), conn(8*n_elem), GL(3*n_node))
  read(u) coord
  read(u) conn
  GL = 0.d0
  do il = 1, n_load
    read(u) ltype, n_el
    read(u) params, rho
    allocate(elems(n_el))
    read(u) elems
    do k = 1, n_el
      icel = elems(k)
      do j = 1, 8
        nod(j) = conn(8*(icel-1)+j)
        xx(j) = coord(3*nod(j)-2); yy(j) = coord(3*nod(j)-1); zz(j) = coord(3*nod(j))
      enddo
      if (ltype == -1) then
        call initMaterial(matl)
        matl%mtype = ELASTIC
        matl%variables(M_YOUNGS) = params(3); matl%variables(M_POISSON) = params(4)
        matl%variables(M_EXAPNSION) = params(2)
        do j = 1, 8
          gausses(j)%pMaterial => matl
          call fstr_init_gauss(gausses(j))
        enddo
        tt = params(0); t0 = params(1); coords = 0.d0
        reftemp_store = params(5)        ! !REFTEMP (m_fstr.f90:115: a pointer into fstr_param)
        REF_TEMP => reftemp_store
        call TLOAD_C3D8IC(361, 8, xx, yy, zz, tt, t0, gausses, vect, 0, coords)
      else
        call DL_C3(361, 8, xx, yy, zz, rho, ltype, params, vect, nsize)
      endif
      do j = 1, 8
        do i = 1, 3
          GL(3*(nod(j)-1)+i) = GL(3*(nod(j)-1)+i) + vect(3*(j-1)+i)
        enddo
      enddo
    enddo
    deallocate(elems)
  enddo
  close(u)
  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) GL
  close(u)
end program ref_load
