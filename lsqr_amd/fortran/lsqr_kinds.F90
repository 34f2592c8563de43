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
!!   -DREAL128             wp = real128 (round 5): the HOST path in full -- `lsqr_solver` with a user `aprod`, its
!!                         `lsqr`, `acheck`, `xcheck`, and `lsqpblas_module` -- in binary128, the reference's
!!                         arithmetic operation for operation (src/lsqr_kinds.F90:20-21; the 18-problem log is byte-
!!                         identical to the reference's -DREAL128 build: tests/test_reference_programs_unchanged.py).
!!                         The device has no binary128 arithmetic: `lsqr_solver_ez` compiles (so that user code does)
!!                         and its `initialize` stops with a message instead of silently computing in binary64.
module lsqr_kinds
   use, intrinsic :: iso_fortran_env, only: real32, real64, real128
   implicit none
   private
#if defined(REAL128)
   integer, parameter, public :: wp = real128
#elif defined(REAL32)
   integer, parameter, public :: wp = real32
#else
   integer, parameter, public :: wp = real64
#endif
   real(wp), parameter, public :: zero = 0.0_wp
   real(wp), parameter, public :: one = 1.0_wp
end module lsqr_kinds
