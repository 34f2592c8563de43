!> EZ-path checks through the Fortran host layer -> C-ABI -> HIP kernels (needs an MI355X).
!!
!! Mirrors what the reference's own EZ test asserts (test/lsqrtest_ez.f90:50,102:
!! max|A x - b| <= 1e-12 on the 3x3 README system and a 3x4 under-determined one), then
!! prints machine-readable results of a generated Poisson system for tests/test_fortran.py
!! to compare against the CPU oracle.
program test_ez
   use lsqr_kinds
   use lsqr_module, only: lsqr_solver_ez
   use iso_fortran_env, only: output_unit
   implicit none
   integer :: nfail
   nfail = 0
   call toy(3, 3, real([1, 4, 7, 2, 5, 88, 3, 66, 9], wp), 'readme_3x3')
   call toy(3, 4, [4.1_wp, 1.1_wp, 11.1_wp, 5.1_wp, -3.1_wp, 3.1_wp, 66.1_wp, 8.1_wp, -87.1_wp, &
                   0.1_wp, -9.1_wp, 2.1_wp], 'ez_3x4')
   call options_and_reuse()
   call poisson(64, 64, 60)
   if (nfail /= 0) error stop 'TEST FAILED'
   write (*, '(A)') 'EZ TESTS PASSED'
contains

   !> dense m-by-n system stored column by column as COO, b = (1,2,3)
   subroutine toy(m, n, a, name)
      integer, intent(in) :: m, n
      real(wp), intent(in) :: a(m*n)
      character(len=*), intent(in) :: name
      integer :: irow(m*n), icol(m*n), i, j, istop
      real(wp) :: b(m), x(n), res(m), amat(m, n)
      type(lsqr_solver_ez) :: solver
      do j = 1, n
         do i = 1, m
            irow((j - 1)*m + i) = i
            icol((j - 1)*m + i) = j
         end do
      end do
      b = [(real(i, wp), i=1, m)]
      amat = reshape(a, [m, n])
      call solver%initialize(m, n, a, irow, icol, itnlim=100, nout=output_unit)
      call solver%solve(b, zero, x, istop)
      res = matmul(amat, x) - b
      write (*, '(A,A,A,I2,A,1P,*(E25.17))') 'TOY ', name, ' istop=', istop, ' x=', x
      write (*, '(A,1P,E10.2)') '  max|Ax-b| =', maxval(abs(res))
      if (istop /= 1 .or. any(abs(res) > 1.0e-12_wp)) nfail = nfail + 1
   end subroutine toy

   !> optional outputs, se, re-initialisation (intent(out) resets options), aprod, assignment
   subroutine options_and_reuse()
      integer, parameter :: m = 3, n = 3
      integer :: irow(9), icol(9), istop, itn, i, j
      real(wp) :: a(9), b(3), x(3), se(3), anorm, acond, rnorm, arnorm, xnorm, xv(3), yv(3)
      type(lsqr_solver_ez) :: s, t
      a = real([1, 4, 7, 2, 5, 88, 3, 66, 9], wp)
      do j = 1, n
         do i = 1, m
            irow((j - 1)*m + i) = i
            icol((j - 1)*m + i) = j
         end do
      end do
      b = [1.0_wp, 2.0_wp, 3.0_wp]
      call s%initialize(m, n, a, irow, icol, atol=1.0e-12_wp, btol=1.0e-12_wp, itnlim=50)
      call s%solve(b, 0.5_wp, x, istop, se=se, itn=itn, anorm=anorm, acond=acond, rnorm=rnorm, arnorm=arnorm, &
                   xnorm=xnorm)
      write (*, '(A,I2,A,I3,1P,A,*(E25.17))') 'DAMPED istop=', istop, ' itn=', itn, ' x=', x
      write (*, '(A,1P,*(E25.17))') 'DAMPED rnorm,xnorm=', rnorm, xnorm
      if (istop /= 3) nfail = nfail + 1
      if (any(se <= zero)) nfail = nfail + 1
      ! aprod: y = y + A x with x = e1 gives column 1
      xv = [1.0_wp, 0.0_wp, 0.0_wp]
      yv = 0.0_wp
      call s%aprod(1, m, n, xv, yv)
      if (any(yv /= [1.0_wp, 4.0_wp, 7.0_wp])) nfail = nfail + 1
      yv = [0.0_wp, 1.0_wp, 0.0_wp]
      xv = 0.0_wp
      call s%aprod(2, m, n, xv, yv)
      if (any(xv /= [4.0_wp, 5.0_wp, 66.0_wp])) nfail = nfail + 1
      ! assignment shares the device matrix; both objects stay usable, either may die first
      t = s
      call s%initialize(m, n, a, irow, icol)          ! defaults are back: atol=btol=0, itnlim=100
      call s%solve(b, zero, x, istop)
      if (istop /= 1) nfail = nfail + 1
      call t%solve(b, 0.5_wp, x, istop)
      if (istop /= 3) nfail = nfail + 1
      write (*, '(A)') 'OPTIONS OK'
   end subroutine options_and_reuse

   !> 5-point Poisson on an nx-by-ny grid, the generator of lsqr_amd/problems.py:poisson2d
   subroutine poisson(nx, ny, itnlim)
      integer, intent(in) :: nx, ny, itnlim
      integer, allocatable :: irow(:), icol(:)
      real(wp), allocatable :: a(:), b(:), x(:)
      integer :: nn, k, i, j, p, istop, itn
      real(wp) :: anorm, rnorm, xnorm
      type(lsqr_solver_ez) :: solver
      nn = nx*ny
      allocate (irow(5*nn), icol(5*nn), a(5*nn), b(nn), x(nn))
      p = 0
      do k = 1, nn
         i = mod(k - 1, nx)
         j = (k - 1)/nx
         if (j > 0) call put(k, k - nx, -one)
         if (i > 0) call put(k, k - 1, -one)
         call put(k, k, 4.0_wp)
         if (i < nx - 1) call put(k, k + 1, -one)
         if (j < ny - 1) call put(k, k + nx, -one)
         b(k) = sin(0.001_wp*real(k, wp))
      end do
      call solver%initialize(nn, nn, a(1:p), irow(1:p), icol(1:p), itnlim=itnlim)
      call solver%solve(b, zero, x, istop, itn=itn, anorm=anorm, rnorm=rnorm, xnorm=xnorm)
      write (*, '(A,I0,A,I0,A,I0,A,I0,A,I0)') 'POISSON nx=', nx, ' ny=', ny, ' nnz=', p, ' istop=', istop, ' itn=', itn
      write (*, '(A,1P,3E25.17)') 'POISSON anorm,rnorm,xnorm=', anorm, rnorm, xnorm
      write (*, '(A,1P,3E25.17)') 'POISSON x(1),x(n/2),x(n)=', x(1), x(nn/2), x(nn)
      if (istop /= 5 .or. itn /= itnlim) nfail = nfail + 1
   contains
      subroutine put(r, c, v)
         integer, intent(in) :: r, c
         real(wp), intent(in) :: v
         p = p + 1
         irow(p) = r
         icol(p) = c
         a(p) = v
      end subroutine put
   end subroutine poisson
end program test_ez
