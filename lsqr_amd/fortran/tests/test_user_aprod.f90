!> Host path: a user type extends lsqr_solver with its own aprod (a dense operator), as the
!! reference's test_solver does (test/lsqrtest_module.f90:35-44, 283-309).  acheck, lsqr and
!! xcheck run on the host around the callback.  Needs no GPU.  Prints results for
!! tests/test_fortran.py, which compares them with the reference on the same matrix.
module dense_operator
   use lsqr_kinds
   use lsqr_module, only: lsqr_solver
   implicit none
   type, extends(lsqr_solver) :: dense_solver
      real(wp), allocatable :: amat(:, :)
   contains
      procedure :: aprod => dense_aprod
   end type dense_solver
contains
   subroutine dense_aprod(me, mode, m, n, x, y)
      class(dense_solver), intent(inout) :: me
      integer, intent(in) :: mode, m, n
      real(wp), dimension(:), intent(inout) :: x
      real(wp), dimension(:), intent(inout) :: y
      if (mode == 1) then
         y(1:m) = y(1:m) + matmul(me%amat, x(1:n))
      else
         x(1:n) = x(1:n) + matmul(transpose(me%amat), y(1:m))
      end if
   end subroutine dense_aprod
end module dense_operator

program test_user_aprod
   use lsqr_kinds
   use dense_operator
   implicit none
   integer, parameter :: m = 40, n = 25
   type(dense_solver) :: s
   real(wp) :: b(m), u(m), v(n), w(n), x(n), se(n), y(m), wm(m), vv(n), xx(n)
   real(wp) :: anorm, acond, rnorm, arnorm, xnorm, t1, t2, t3, damp
   integer :: i, j, istop, itn, inform
   allocate (s%amat(m, n))
   do j = 1, n
      do i = 1, m
         s%amat(i, j) = one/real(i + 2*j + 1, wp)          ! exact rational entries
         if (i == j) s%amat(i, j) = s%amat(i, j) + 2.0_wp
      end do
   end do
   b = [(one/real(i, wp) - 0.25_wp, i=1, m)]
   damp = 0.125_wp
   call s%acheck(m, n, 0, epsilon(one), vv, wm, xx, y, inform)
   write (*, '(A,I2)') 'ACHECK inform=', inform
   u = b
   call s%lsqr(m, n, damp, .true., u, v, w, x, se, 1.0e-12_wp, 1.0e-12_wp, 1.0e8_wp, 200, 0, &
               istop, itn, anorm, acond, rnorm, arnorm, xnorm)
   write (*, '(A,I2,A,I4)') 'LSQR istop=', istop, ' itn=', itn
   write (*, '(A,1P,5E25.17)') 'LSQR norms=', anorm, acond, rnorm, arnorm, xnorm
   write (*, '(A,1P,*(E25.17))') 'LSQR x=', x
   write (*, '(A,1P,*(E25.17))') 'LSQR se=', se
   call s%xcheck(m, n, 0, anorm, damp, epsilon(one), b, u, v, w, x, inform, t1, t2, t3)
   write (*, '(A,I2,1P,3E12.4)') 'XCHECK inform,tests=', inform, t1, t2, t3
   if (inform /= 3 .and. inform /= 2 .and. inform /= 1) error stop 'TEST FAILED'
   write (*, '(A)') 'USER APROD TESTS PASSED'
end program test_user_aprod
