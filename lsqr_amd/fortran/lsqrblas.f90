!> Host BLAS-1 with the reference's public interface (`lsqpblas_module` -- the module
!! name really is spelt this way in the reference, src/lsqrblas.f90:8,16).
!!
!! These act on HOST arrays handed in by user code (e.g. a user `aprod`); inside the
!! GPU-resident iteration the same operations are fused into the HIP kernels
!! (lsqr_amd/csrc/vec.h).  Written from the BLAS-1 specification, not from the
!! reference's unrolled loops.
module lsqpblas_module
   use lsqr_kinds, only: wp, zero, one
   implicit none
   private
   public :: dcopy, ddot, dnrm2, dscal
contains

   !> y <- x                                   (replaces src/lsqrblas.f90:25-67)
   subroutine dcopy(n, dx, incx, dy, incy)
      integer :: n, incx, incy
      real(wp) :: dx(*), dy(*)
      integer :: k, ix, iy
      if (n <= 0) return
      ix = first_index(n, incx)
      iy = first_index(n, incy)
      do k = 1, n
         dy(iy) = dx(ix)
         ix = ix + incx
         iy = iy + incy
      end do
   end subroutine dcopy

   !> x . y                                    (replaces src/lsqrblas.f90:74-116)
   real(wp) function ddot(n, dx, incx, dy, incy)
      integer :: n, incx, incy
      real(wp) :: dx(*), dy(*)
      integer :: k, ix, iy
      real(wp) :: acc
      acc = zero
      if (n > 0) then
         ix = first_index(n, incx)
         iy = first_index(n, incy)
         do k = 1, n
            acc = acc + dx(ix)*dy(iy)
            ix = ix + incx
            iy = iy + incy
         end do
      end if
      ddot = acc
   end function ddot

   !> ||x||_2 without overflow / underflow     (replaces src/lsqrblas.f90:123-159)
   !! One pass, LAPACK's dlassq recurrence: the running pair (big, acc) holds sum x_k^2 = big^2 * acc with big the
   !! largest magnitude met so far; a new largest rescales acc, anything else adds (|x|/big)^2.  The reference does
   !! exactly this (src/lsqrblas.f90:143-154), and the host lsqr / acheck / xcheck of a user type that supplies its own
   !! aprod take every beta and alpha from here: the SAME recurrence, operation for operation, is what makes that path
   !! bit-identical to the reference's (tests/test_reference_programs_unchanged.py compares the 18-problem log).
   !! (Round 4 took two passes -- max, then sum (x/max)^2 -- which is as accurate and differs in the last bits.)
   real(wp) function dnrm2(n, x, incx)
      integer :: n, incx
      real(wp) :: x(*)
      integer :: k, ix
      real(wp) :: big, acc, mag
      dnrm2 = zero
      if (n < 1 .or. incx < 1) return
      if (n == 1) then
         dnrm2 = abs(x(1))
         return
      end if
      big = zero
      acc = one
      ix = 1
      do k = 1, n
         mag = abs(x(ix))
         ix = ix + incx
         if (mag == zero) cycle
         if (mag > big) then
            acc = one + acc*(big/mag)**2
            big = mag
         else
            acc = acc + (mag/big)**2
         end if
      end do
      dnrm2 = big*sqrt(acc)
   end function dnrm2

   !> x <- a x                                 (replaces src/lsqrblas.f90:166-201)
   subroutine dscal(n, da, dx, incx)
      integer :: n, incx
      real(wp) :: da, dx(*)
      integer :: k, ix
      if (n <= 0 .or. incx <= 0) return
      ix = 1
      do k = 1, n
         dx(ix) = da*dx(ix)
         ix = ix + incx
      end do
   end subroutine dscal

   pure integer function first_index(n, inc)
      integer, intent(in) :: n, inc
      first_index = 1
      if (inc < 0) first_index = (1 - n)*inc + 1
   end function first_index

end module lsqpblas_module
