!> `module m_fstr_StiffMatrix` of a GPU build of fistr1 (INTEGRATION.md section 5): same module and procedure name as
!> fistr1/src/analysis/static/fstr_StiffMatrix.f90:7-18.  The reference's module is kept in the binary under the name
!> m_fstr_StiffMatrix_ref (one renamed line) and serves every deck the device kernels do not cover.
module m_fstr_StiffMatrix
  use m_fstr
  use hecmw, only: hecmw_Wtime
  use m_fstr_StiffMatrix_ref, only: fstr_StiffMatrix_ref => fstr_StiffMatrix
  use fstr_device_hip
  implicit none
  private
  public :: fstr_StiffMatrix
contains
  subroutine fstr_StiffMatrix(hecMESH, hecMAT, fstrSOLID, time, tincr)
    type (hecmwST_local_mesh)  :: hecMESH
    type (hecmwST_matrix)      :: hecMAT
    type (fstr_solid)          :: fstrSOLID
    real(kind=kreal), intent(in) :: time
    real(kind=kreal), intent(in) :: tincr
    real(kind=kreal) :: t0
    t0 = hecmw_Wtime()
    if (fsd_stiffness(hecMESH, hecMAT, fstrSOLID)) then        ! tangent assembled on the device; D / AL / AU of hecMAT are not touched
      call fsd_report('fstr_StiffMatrix on the device', hecmw_Wtime() - t0)
      return
    endif
    call fstr_StiffMatrix_ref(hecMESH, hecMAT, fstrSOLID, time, tincr)
    call fsd_report('fstr_StiffMatrix on the host', hecmw_Wtime() - t0)
  end subroutine fstr_StiffMatrix
end module m_fstr_StiffMatrix
