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
   !! Two passes: the largest magnitude, then the sum of squares of x/max.
   real(wp) function dnrm2(n, x, incx)
      integer :: n, incx
      real(wp) :: x(*)
      integer :: k, ix
      real(wp) :: big, acc, t
      dnrm2 = zero
      if (n < 1 .or. incx < 1) return
      if (n == 1) then
         dnrm2 = abs(x(1))
         return
      end if
      big = zero
      ix = 1
      do k = 1, n
         big = max(big, abs(x(ix)))
         ix = ix + incx
      end do
      if (big == zero) return
      acc = zero
      ix = 1
      do k = 1, n
         t = x(ix)/big
         acc = acc + t*t
         ix = ix + incx
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
