!> `module m_fstr_Cutback` of a GPU build of fistr1 (INTEGRATION.md section 5): same module and procedure names as
!> fistr1/src/analysis/static/fstr_Cutback.f90:7-198.  The reference's module stays in the binary as m_fstr_Cutback_ref and does
!> all the host-side work (unode, QFORCE, fstrSOLID%elements(:)%gausses(:), contact states); while the quadrature-point history of
!> the run lives on the device (fstr_device_hip) its copy there is saved / rolled back too, so that an `!AUTOINC_PARAM` / cutback deck
!> keeps the element loops on the device (fstr_solve_NLGEOM.f90:156-197 calls save after every converged sub-step, load after a
!> failed one).
module m_fstr_Cutback
  use m_fstr
  use m_fstr_Cutback_ref, only: fstr_cutback_active_ref => fstr_cutback_active, fstr_cutback_init_ref => fstr_cutback_init, &
    fstr_cutback_finalize_ref => fstr_cutback_finalize, fstr_cutback_save_ref => fstr_cutback_save, &
    fstr_cutback_load_ref => fstr_cutback_load
  use fstr_device_hip, only: fsd_cutback
  implicit none
  private
  public :: fstr_cutback_active, fstr_cutback_init, fstr_cutback_finalize, fstr_cutback_save, fstr_cutback_load
contains
  logical function fstr_cutback_active()
    fstr_cutback_active = fstr_cutback_active_ref()
  end function fstr_cutback_active

  subroutine fstr_cutback_init(hecMESH, fstrSOLID, fstrPARAM)
    type(hecmwST_local_mesh) :: hecMESH
    type(fstr_param)         :: fstrPARAM
    type(fstr_solid)         :: fstrSOLID
    call fstr_cutback_init_ref(hecMESH, fstrSOLID, fstrPARAM)
  end subroutine fstr_cutback_init

  subroutine fstr_cutback_finalize(fstrSOLID)
    type(fstr_solid) :: fstrSOLID
    call fstr_cutback_finalize_ref(fstrSOLID)
  end subroutine fstr_cutback_finalize

  subroutine fstr_cutback_save(fstrSOLID, infoCTChange, infoCTChange_bak)
    type(fstr_solid), intent(inout)              :: fstrSOLID
    type(fstr_info_contactChange), intent(inout) :: infoCTChange
    type(fstr_info_contactChange), intent(inout) :: infoCTChange_bak
    call fstr_cutback_save_ref(fstrSOLID, infoCTChange, infoCTChange_bak)
    if (fstr_cutback_active_ref()) call fsd_cutback(0)
  end subroutine fstr_cutback_save

  subroutine fstr_cutback_load(fstrSOLID, infoCTChange, infoCTChange_bak)
    type(fstr_solid), intent(inout)              :: fstrSOLID
    type(fstr_info_contactChange), intent(inout) :: infoCTChange
    type(fstr_info_contactChange), intent(inout) :: infoCTChange_bak
    call fstr_cutback_load_ref(fstrSOLID, infoCTChange, infoCTChange_bak)
    if (fstr_cutback_active_ref()) call fsd_cutback(1)
  end subroutine fstr_cutback_load
end module m_fstr_Cutback
