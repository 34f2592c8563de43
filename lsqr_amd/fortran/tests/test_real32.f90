!> The REAL32 build of the host layer (-DREAL32: wp = real32, as src/lsqr_kinds.F90:16-17 of the
!! reference): user arrays and scalars are real32 and so is everything stored on the device (matrix
!! values and u, v, w, x, se); arithmetic in registers is binary64.  LSQRHIP_REAL32_MIXED=1 in the
!! environment: binary64 storage on the device, real32 at the boundary only.  Prints results for
!! tests/test_fortran.py.
program test_real32
   use lsqr_kinds
   use lsqr_module, only: lsqr_solver_ez
   implicit none
   integer, parameter :: nx = 40, ny = 30, n = nx*ny
   integer, allocatable :: irow(:), icol(:)
   real(wp), allocatable :: a(:), b(:), x(:), se(:)
   real(wp) :: anorm, acond, rnorm, arnorm, xnorm
   integer :: i, j, k, nnz, istop, itn, ngpu
   character(len=32) :: arg
   type(lsqr_solver_ez) :: s

   if (kind(one) /= kind(1.0)) error stop 'TEST FAILED: this program must be built with -DREAL32'
   ! usage: test_real32 [ngpu]   -- with ngpu: the same systems with their rows sharded over ngpu devices
   ngpu = 0
   if (command_argument_count() >= 1) then
      call get_command_argument(1, arg)
      read (arg, *) ngpu
   end if
   ! README system (README.md:33-38 of the reference) in real32
   if (ngpu > 0) then
      call s%initialize(3, 3, real([1, 4, 7, 2, 5, 88, 3, 66, 9], wp), [1, 2, 3, 1, 2, 3, 1, 2, 3], &
                        [1, 1, 1, 2, 2, 2, 3, 3, 3], ngpu=ngpu)
   else
      call s%initialize(3, 3, real([1, 4, 7, 2, 5, 88, 3, 66, 9], wp), [1, 2, 3, 1, 2, 3, 1, 2, 3], &
                        [1, 1, 1, 2, 2, 2, 3, 3, 3])
   end if
   allocate (x(3))
   call s%solve(real([1, 2, 3], wp), zero, x, istop)
   write (*, '(A,I2,1P,3E16.8)') 'README32 istop,x=', istop, x
   if (istop /= 1) error stop 'TEST FAILED'
   if (any(abs(x - [1.242424_wp, -6.060606e-2_wp, -4.040404e-2_wp]) > 5.0e-6_wp)) error stop 'TEST FAILED'
   ! aprod in real32 (mode 1: y + A x; mode 2: x + A' y): small integers, exact in any precision
   block
      real(wp) :: xx(3), yy(3)
      xx = real([1, 2, 3], wp); yy = real([1, 1, 1], wp)
      call s%aprod(1, 3, 3, xx, yy)
      if (any(yy /= real([1 + 1 + 4 + 9, 1 + 4 + 10 + 198, 1 + 7 + 176 + 27], wp))) error stop 'TEST FAILED: aprod mode 1'
      xx = real([1, 2, 3], wp); yy = real([1, 0, 2], wp)
      call s%aprod(2, 3, 3, xx, yy)
      if (any(xx /= real([1 + 1 + 14, 2 + 2 + 176, 3 + 3 + 18], wp))) error stop 'TEST FAILED: aprod mode 2'
   end block
   deallocate (x)

   ! 5-point stencil with non-integer coefficients, damped, with standard errors
   allocate (irow(5*n), icol(5*n), a(5*n), b(n), x(n), se(n))
   nnz = 0
   do j = 1, ny
      do i = 1, nx
         k = (j - 1)*nx + i
         call put(k, k, 4.25_wp)
         if (i > 1) call put(k, k - 1, -1.125_wp)
         if (i < nx) call put(k, k + 1, -0.875_wp)
         if (j > 1) call put(k, k - nx, -1.0625_wp)
         if (j < ny) call put(k, k + nx, -0.9375_wp)
         b(k) = real(mod(7*k, 13), wp)*0.25_wp - 1.5_wp
      end do
   end do
   if (ngpu > 0) then
      call s%initialize(n, n, a(1:nnz), irow(1:nnz), icol(1:nnz), atol=1.0e-7_wp, btol=1.0e-7_wp, itnlim=500, ngpu=ngpu)
   else
      call s%initialize(n, n, a(1:nnz), irow(1:nnz), icol(1:nnz), atol=1.0e-7_wp, btol=1.0e-7_wp, itnlim=500)
   end if
   call s%solve(b, 0.0625_wp, x, istop, se=se, itn=itn, anorm=anorm, acond=acond, rnorm=rnorm, arnorm=arnorm, &
                xnorm=xnorm)
   write (*, '(A,I2,A,I4)') 'STENCIL32 istop=', istop, ' itn=', itn
   write (*, '(A,1P,5E16.8)') 'STENCIL32 norms=', anorm, acond, rnorm, arnorm, xnorm
   write (*, '(A,1P,*(E16.8))') 'STENCIL32 x=', x
   write (*, '(A,1P,*(E16.8))') 'STENCIL32 se=', se(1:8)
   write (*, '(A)') 'REAL32 TESTS PASSED'
contains
   subroutine put(r, c, v)
      integer, intent(in) :: r, c
      real(wp), intent(in) :: v
      nnz = nnz + 1
      irow(nnz) = r
      icol(nnz) = c
      a(nnz) = v
   end subroutine put
end program test_real32
