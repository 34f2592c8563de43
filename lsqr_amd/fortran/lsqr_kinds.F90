!> Kinds and constants of the LSQR host layer.
!!
!! Same public names as the reference's `lsqr_kinds` (src/lsqr_kinds.F90:16-28) so that
!! user code `use lsqr_kinds` keeps compiling.  The device path computes in IEEE
!! binary64 only, so `wp` is fixed to real64 (= real(c_double)); the REAL32 / REAL128
!! builds of the reference are out of scope (SURVEY.md section 8f, rank 4).
module lsqr_kinds
   use, intrinsic :: iso_fortran_env, only: real64
   implicit none
   private
#if defined(REAL32) || defined(REAL128)
#error "lsqr-mi355x: only the default REAL64 build exists on the device path"
#endif
   integer, parameter, public :: wp = real64
   real(wp), parameter, public :: zero = 0.0_wp
   real(wp), parameter, public :: one = 1.0_wp
end module lsqr_kinds
