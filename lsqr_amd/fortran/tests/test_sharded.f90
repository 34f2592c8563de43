!> `initialize(..., ngpu=N)`: the EZ class with its rows sharded over N GPUs of the node (RCCL inside
!! the library, csrc/shard_engine.h).  usage: test_sharded [ngpu]   (default 1; a 1-GPU box runs 1)
!!
!! Same checks as the reference's EZ test (test/lsqrtest_ez.f90:50: max|A x - b| <= 1e-12 on the README
!! system), then a generated Poisson system printed for tests/test_fortran.py to compare with the oracle.
!! With more GPUs requested than the node has, `initialize` must `error stop` (never run on fewer silently).
program test_sharded
   use lsqr_kinds
   use lsqr_module, only: lsqr_solver_ez
   implicit none
   integer :: ngpu, nfail
   character(len=32) :: arg
   ngpu = 1
   if (command_argument_count() >= 1) then
      call get_command_argument(1, arg)
      read (arg, *) ngpu
   end if
   nfail = 0
   call readme()
   call poisson(64, 64, 60)
   if (nfail /= 0) error stop 'TEST FAILED'
   write (*, '(A,I0)') 'SHARDED TESTS PASSED ngpu=', ngpu
contains

   subroutine readme()
      integer, parameter :: m = 3, n = 3
      integer :: irow(9), icol(9), i, j, istop
      real(wp) :: a(9), b(3), x(3), amat(3, 3), xv(3), yv(3)
      type(lsqr_solver_ez) :: solver
      a = real([1, 4, 7, 2, 5, 88, 3, 66, 9], wp)
      do j = 1, n
         do i = 1, m
            irow((j - 1)*m + i) = i
            icol((j - 1)*m + i) = j
         end do
      end do
      b = [1.0_wp, 2.0_wp, 3.0_wp]
      amat = reshape(a, [m, n])
      call solver%initialize(m, n, a, irow, icol, itnlim=100, ngpu=ngpu)
      call solver%solve(b, zero, x, istop)
      write (*, '(A,I2,A,1P,3E25.17)') 'README istop=', istop, ' x=', x
      if (istop /= 1 .or. any(abs(matmul(amat, x) - b) > 1.0e-12_wp)) nfail = nfail + 1
      xv = [1.0_wp, -2.0_wp, 0.5_wp]          ! aprod through the sharded handle
      yv = zero
      call solver%aprod(1, m, n, xv, yv)
      if (any(abs(yv - matmul(amat, xv)) > 1.0e-12_wp)) nfail = nfail + 1
   end subroutine readme

   !> 5-point Laplacian on an nx-by-ny grid, b_k = sin(0.001 k) (lsqr_amd/problems.py poisson2d)
   subroutine poisson(nx, ny, itnlim)
      integer, intent(in) :: nx, ny, itnlim
      integer, allocatable :: irow(:), icol(:)
      real(wp), allocatable :: a(:), b(:), x(:)
      integer :: n, k, i, j, nz, istop, itn
      real(wp) :: anorm, rnorm
      type(lsqr_solver_ez) :: solver
      n = nx*ny
      allocate (irow(5*n), icol(5*n), a(5*n), b(n), x(n))
      nz = 0
      do j = 1, ny
         do i = 1, nx
            k = (j - 1)*nx + i
            if (j > 1) call put(k, k - nx, -1.0_wp)
            if (i > 1) call put(k, k - 1, -1.0_wp)
            call put(k, k, 4.0_wp)
            if (i < nx) call put(k, k + 1, -1.0_wp)
            if (j < ny) call put(k, k + nx, -1.0_wp)
            b(k) = sin(0.001_wp*real(k, wp))
         end do
      end do
      call solver%initialize(n, n, a(1:nz), irow(1:nz), icol(1:nz), itnlim=itnlim, ngpu=ngpu)
      call solver%solve(b, zero, x, istop, itn=itn, anorm=anorm, rnorm=rnorm)
      write (*, '(A,I0,A,I0,A,I0,A,I0)') 'POISSON nx=', nx, ' ny=', ny, ' istop=', istop, ' itn=', itn
      write (*, '(A,1P,2E25.17)') 'POISSON_NORMS ', anorm, rnorm
      write (*, '(A,1P,4E25.17)') 'POISSON_X ', x(1), x(n/3), x(n/2), x(n)
   contains
      subroutine put(r, c, v)
         integer, intent(in) :: r, c
         real(wp), intent(in) :: v
         nz = nz + 1
         irow(nz) = r; icol(nz) = c; a(nz) = v
      end subroutine put
   end subroutine poisson
end program test_sharded
