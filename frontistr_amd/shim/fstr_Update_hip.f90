!> `module m_fstr_Update` of a GPU build of fistr1 (INTEGRATION.md section 5): same module and procedure names as
!> fistr1/src/analysis/static/fstr_Update.f90:6, :25, :297.  The reference's module stays in the binary as m_fstr_Update_ref.
module m_fstr_Update
  use m_fstr
  use hecmw, only: hecmw_Wtime
  use m_fstr_Update_ref, only: fstr_UpdateNewton_ref => fstr_UpdateNewton, fstr_UpdateState_ref => fstr_UpdateState
  use fstr_device_hip
  implicit none
  private
  public :: fstr_UpdateNewton, fstr_UpdateState
contains
  subroutine fstr_UpdateNewton(hecMESH, hecMAT, fstrSOLID, time, tincr, iter, strainEnergy)
    type (hecmwST_matrix)       :: hecMAT
    type (hecmwST_local_mesh)   :: hecMESH
    type (fstr_solid)           :: fstrSOLID
    real(kind=kreal), intent(in) :: time
    real(kind=kreal), intent(in) :: tincr
    integer, intent(in)         :: iter
    real(kind=kreal), optional :: strainEnergy
    real(kind=kreal) :: t0
    if (present(strainEnergy)) then
      call fstr_UpdateNewton_ref(hecMESH, hecMAT, fstrSOLID, time, tincr, iter, strainEnergy)
      return
    endif
    t0 = hecmw_Wtime()
    if (fsd_update_newton(hecMESH, fstrSOLID)) then
      call fsd_report('fstr_UpdateNewton on the device', hecmw_Wtime() - t0)
      return
    endif
    call fstr_UpdateNewton_ref(hecMESH, hecMAT, fstrSOLID, time, tincr, iter)
    call fsd_report('fstr_UpdateNewton on the host', hecmw_Wtime() - t0)
  end subroutine fstr_UpdateNewton

  subroutine fstr_UpdateState(hecMESH, fstrSOLID, tincr)
    type(hecmwST_local_mesh) :: hecMESH
    type(fstr_solid) :: fstrSOLID
    real(kind=kreal) :: tincr
    real(kind=kreal) :: t0
    t0 = hecmw_Wtime()
    if (fsd_update_state(hecMESH, fstrSOLID)) then
      call fsd_report('fstr_UpdateState on the device', hecmw_Wtime() - t0)
      return
    endif
    call fstr_UpdateState_ref(hecMESH, fstrSOLID, tincr)
    call fsd_report('fstr_UpdateState on the host', hecmw_Wtime() - t0)
  end subroutine fstr_UpdateState
end module m_fstr_Update
