!> Kinds and constants of the LSQR host layer.
!!
!! Same public names as the reference's `lsqr_kinds` (src/lsqr_kinds.F90:16-28) so that
!! user code `use lsqr_kinds` keeps compiling, including the reference's precision macro:
!!
!!   (default), -DREAL64   wp = real64: host arrays are handed to the device as they are
!!   -DREAL32              wp = real32: a MIXED-precision build (SURVEY.md 8f rank 4) -- the
!!                         user's arrays and scalars are real32 like the reference's REAL32
!!                         build (src/lsqr_kinds.F90:16-17), the device computes in binary64
!!                         on the exactly converted values and results are rounded to real32
!!                         on the way out; so x is at least as accurate as the reference's
!!                         all-real32 iteration
!!   -DREAL128             not offered: the device has no binary128 arithmetic
module lsqr_kinds
   use, intrinsic :: iso_fortran_env, only: real32, real64
   implicit none
   private
#if defined(REAL128)
#error "lsqr-mi355x: no REAL128 build (the device path computes in binary64)"
#endif
#if defined(REAL32)
   integer, parameter, public :: wp = real32
#else
   integer, parameter, public :: wp = real64
#endif
   real(wp), parameter, public :: zero = 0.0_wp
   real(wp), parameter, public :: one = 1.0_wp
end module lsqr_kinds
