!> TEST INFRASTRUCTURE ONLY (oracle).  Our own driver around the *reference's*
!> solver modules: reads a flat binary system, fills hecmwST_matrix /
!> hecmwST_local_mesh exactly as fistr1 would (m_fstr.f90:807-857 allocates the
!> same members) and calls the reference entry points
!>   mode 1: hecmw_solve_iterative   (hecmw_solver_Iterative.f90:13)
!>   mode 2: hecmw_matvec            (hecmw_solver_las.f90:57)
!>   mode 3: hecmw_precond_setup + hecmw_precond_apply (hecmw_precond.f90:28,75)
!>   mode 5: hecmw_matvec, hecmw_solve of a scaled matrix, hecmw_matvec again (out: the second product)
!> stdout carries the reference's own ITERLOG / summary lines.
!>
!> usage: ref_solve in.bin out.bin
!> in.bin  (little endian, stream):
!>   int32  magic(=1179210580) mode N NP NPL NPU nrepeat
!>   int32  Iarray(100);  real64 Rarray(100)
!>   int32  indexL(0:NP) indexU(0:NP) itemL(NPL) itemU(NPU)
!>   real64 D(9NP) AL(9NPL) AU(9NPU) B(3NP) X(3NP)
!>   optional (a subdomain with halo tables, hecmw_util_f.F90:298-310):
!>   int32  n_neighbor_pe PETOT my_rank; neighbor_pe(n) import_index(0:n) export_index(0:n) import_item(*) export_item(*)
!> out.bin: int32 Iarray(100); real64 Rarray(100); real64 X(3NP) (mode 2/3: Y/Z)
!>          real64 t_total
program ref_solve
  use hecmw_util
  use hecmw_matrix_misc
#ifdef USE_SHIM
  use hecmw_solver            ! frontistr_amd/shim/hecmw_solver_hip.f90: same module / procedure names as the reference
#else
  use hecmw_solver_iterative
#endif
  use hecmw_solver_las
  use hecmw_precond
  implicit none
  type(hecmwST_local_mesh) :: hecMESH
  type(hecmwST_matrix)     :: hecMAT
  character(len=1024) :: fin, fout
  integer(kind=4) :: magic, mode, N, NP, NPL, NPU, nrepeat, u, irep, nd, nnb, petot, myrank, ios
  integer(kind=4) :: Iarr(100)
  real(kind=8)    :: Rarr(100), t0, t1, tcomm
  real(kind=8), allocatable :: Y(:), WK(:), X0(:)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, mode, N, NP, NPL, NPU, nrepeat
  if (magic /= 1179210580) stop 'bad magic'
  nd = 3                          ! mode = 100*NDOF + mode selects another block size (hecMAT%NDOF)
  if (mode >= 100) then
    nd = mode / 100
    mode = mod(mode, 100)
  endif
  read(u) Iarr
  read(u) Rarr

  call hecmw_nullify_mesh(hecMESH)
  call hecmw_mat_init(hecMAT)
  hecMAT%Iarray = Iarr
  hecMAT%Rarray = Rarr
  hecMAT%N = N; hecMAT%NP = NP; hecMAT%NPL = NPL; hecMAT%NPU = NPU; hecMAT%NDOF = nd
  hecMAT%NPCL = 0; hecMAT%NPCU = 0
  allocate(hecMAT%indexL(0:NP), hecMAT%indexU(0:NP), hecMAT%itemL(NPL), hecMAT%itemU(NPU))
  allocate(hecMAT%D(nd*nd*NP), hecMAT%AL(nd*nd*NPL), hecMAT%AU(nd*nd*NPU), hecMAT%B(nd*NP), hecMAT%X(nd*NP))
  read(u) hecMAT%indexL
  read(u) hecMAT%indexU
  read(u) hecMAT%itemL
  read(u) hecMAT%itemU
  read(u) hecMAT%D
  read(u) hecMAT%AL
  read(u) hecMAT%AU
  read(u) hecMAT%B
  read(u) hecMAT%X

  hecMESH%zero = 0; hecMESH%MPI_COMM = 0; hecMESH%PETOT = 1; hecMESH%PEsmpTOT = 1
  hecMESH%my_rank = 0; hecMESH%n_subdomain = 1; hecMESH%n_neighbor_pe = 0
  hecMESH%n_node = NP; hecMESH%nn_internal = N; hecMESH%n_dof = nd
  hecMESH%nn_middle = NP
  hecMESH%mpc%n_mpc = 0
  nnb = 0
  read(u, iostat=ios) nnb, petot, myrank
  if (ios /= 0) nnb = 0
  if (nnb > 0) then
    hecMESH%n_neighbor_pe = nnb; hecMESH%PETOT = petot; hecMESH%my_rank = myrank
    allocate(hecMESH%neighbor_pe(nnb), hecMESH%import_index(0:nnb), hecMESH%export_index(0:nnb))
    read(u) hecMESH%neighbor_pe
    read(u) hecMESH%import_index
    read(u) hecMESH%export_index
    allocate(hecMESH%import_item(hecMESH%import_index(nnb)), hecMESH%export_item(hecMESH%export_index(nnb)))
    read(u) hecMESH%import_item
    read(u) hecMESH%export_item
  else
    allocate(hecMESH%neighbor_pe(0), hecMESH%import_index(0:0), hecMESH%export_index(0:0))
    allocate(hecMESH%import_item(0), hecMESH%export_item(0))
    hecMESH%import_index(0) = 0; hecMESH%export_index(0) = 0
  endif
  close(u)

  allocate(Y(nd*NP), WK(nd*NP), X0(nd*NP))
  X0 = hecMAT%X
  t0 = hecmw_Wtime()
  select case (mode)
  case (1)
    do irep = 1, max(nrepeat, 1)
      if (irep > 1) then
        hecMAT%X = X0
        hecMAT%Iarray = Iarr
      endif
#ifdef USE_SHIM
      call hecmw_solve(hecMESH, hecMAT)        ! exactly what fistr1/src/lib/solve_LINEQ.f90:22 does
#else
      call hecmw_solve_iterative(hecMESH, hecMAT)
#endif
    enddo
    Y = hecMAT%X
  case (4)   ! a sequence of solves with changing values and Iarray(97) = 1: the recycle policy of the preconditioner
    do irep = 1, max(nrepeat, 1)   ! (hecmw_mat_recycle_precond_setting, hecmw_matrix_misc.f90:678-697) as in a Newton loop
      if (irep > 1) then
        hecMAT%D = hecMAT%D * 1.1d0
        hecMAT%X = X0
        hecMAT%Iarray(97) = 1
        hecMAT%Iarray(98) = 0
      endif
#ifdef USE_SHIM
      call hecmw_solve(hecMESH, hecMAT)
#else
      call hecmw_solve_iterative(hecMESH, hecMAT)
#endif
    enddo
    Y = hecMAT%X
  case (2)
    tcomm = 0.d0
    do irep = 1, max(nrepeat, 1)
      call hecmw_matvec(hecMESH, hecMAT, hecMAT%X, Y, tcomm)
    enddo
    Y(nd*N+1:) = hecMAT%X(nd*N+1:)     ! the halo part of X after the update, in the unused tail of Y
  case (5)   ! hecmw_matvec, then a solve of ANOTHER matrix of the same shape (D scaled), then hecmw_matvec with the first one again:
    tcomm = 0.d0   ! the external callers' pattern around a solve (a resident-values binding must notice whose values the device holds)
    call hecmw_matvec(hecMESH, hecMAT, hecMAT%X, Y, tcomm)
    WK = hecMAT%D
    hecMAT%D = hecMAT%D * 2.d0
    X0 = hecMAT%X
#ifdef USE_SHIM
    call hecmw_solve(hecMESH, hecMAT)
#else
    call hecmw_solve_iterative(hecMESH, hecMAT)
#endif
    hecMAT%D = WK(1:size(hecMAT%D))
    hecMAT%X = X0
    call hecmw_matvec(hecMESH, hecMAT, hecMAT%X, Y, tcomm)
    Y(nd*N+1:) = hecMAT%X(nd*N+1:)
  case (3)
    tcomm = 0.d0
    call hecmw_precond_setup(hecMAT, hecMESH, 1)
    do irep = 1, max(nrepeat, 1)
      call hecmw_precond_apply(hecMESH, hecMAT, hecMAT%B, Y, WK, tcomm)
    enddo
  case default
    stop 'bad mode'
  end select
  t1 = hecmw_Wtime()

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) hecMAT%Iarray
  write(u) hecMAT%Rarray
  write(u) Y
  write(u) t1 - t0
  close(u)
end program ref_solve
