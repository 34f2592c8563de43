!> Device operators from Fortran (lsqr_device_module):
!!  1. the reference's test problem P(m,n,nduplc,npower,damp) as the built-in device operator,
!!     through the reference driver's sequence acheck -> lsqr -> xcheck -> error against xtrue
!!     (test/lsqrtest_module.f90:119-272);
!!  2. a USER type that extends lsqr_solver_device and applies its operator with library calls on
!!     the stream it is given (here: another handle's device product) -- same answers as 1.
!! Prints results for tests/test_fortran.py.
module wrapped_operator
   use, intrinsic :: iso_c_binding
   use lsqr_kinds
   use lsqr_device_module
   implicit none
   type, extends(lsqr_solver_device) :: wrapped_solver
      type(c_ptr) :: inner = c_null_ptr     !< handle whose product this operator forwards to
      integer :: calls = 0
   contains
      procedure :: aprod_device => wrapped_aprod
   end type wrapped_solver
contains
   subroutine wrapped_aprod(me, mode, m, n, x, y, stream)
      class(wrapped_solver), intent(inout) :: me
      integer, intent(in) :: mode, m, n
      type(c_ptr), intent(in) :: x, y, stream
      integer(c_int) :: rc
      me%calls = me%calls + 1
      rc = lsqrhip_set_stream(me%inner, stream)
      if (rc == 0) rc = lsqrhip_aprod_device(me%inner, int(mode, c_int), x, y)
      if (rc /= 0) error stop 'inner aprod failed'
      if (m < 0 .or. n < 0) error stop 'unreachable'
   end subroutine wrapped_aprod
end module wrapped_operator

program test_device_operator
   use, intrinsic :: iso_c_binding
   use lsqr_kinds
   use lsqr_device_module
   use wrapped_operator
   implicit none
   ! (the same source serves the REAL32 build -- lib/test_device_operator32, -DREAL32 modules: the operator's vectors
   !  are real32 arrays on the device there --, with the suite's first problem, the one real32 can still solve to the
   !  test's 1e-3: P(2000, 1000, 40, 2, 1e-8); binary64: P(2000, 1000, 40, 3, 1e-9))
   logical, parameter :: single = epsilon(one) > 1.0e-10_wp
   integer, parameter :: m = 2000, n = 1000, nduplc = 40, npower = merge(2, 3, single)
   real(wp), parameter :: damp = merge(1.0e-8_wp, 1.0e-9_wp, single)
   type(lsqr_test_problem_device), target :: p
   type(wrapped_solver), target :: ws
   real(wp) :: u(m), v(n), w(n), x(n), x2(n), se(1), y(m), wm(m), vv(n), xx(n)
   real(wp) :: atol, conlim, anorm, acond, rnorm, arnorm, xnorm, t1, t2, t3, enorm
   integer :: istop, itn, itn2, inform, itnlim

   call p%create(m, n, nduplc, npower, damp)
   write (*, '(A,1P,2E25.17)') 'LSTP acond,rnorm=', p%acond, p%rnorm
   call p%acheck(m, n, 0, epsilon(one), vv, wm, xx, y, inform)
   write (*, '(A,I2)') 'ACHECK inform=', inform
   atol = epsilon(one)**0.99_wp
   conlim = 1000.0_wp*p%acond
   itnlim = 4*(m + n + 50)
   u = p%b
   call p%lsqr(m, n, damp, .false., u, v, w, x, se, atol, atol, conlim, itnlim, 0, &
               istop, itn, anorm, acond, rnorm, arnorm, xnorm)
   write (*, '(A,I2,A,I5)') 'LSQR istop=', istop, ' itn=', itn
   write (*, '(A,1P,5E25.17)') 'LSQR norms=', anorm, acond, rnorm, arnorm, xnorm
   call p%xcheck(m, n, 0, anorm, damp, epsilon(one), p%b, u, v, w, x, inform, t1, t2, t3)
   write (*, '(A,I2,1P,3E12.4)') 'XCHECK inform,tests=', inform, t1, t2, t3
   enorm = sqrt(sum((x - p%xtrue)**2))/(one + sqrt(sum(p%xtrue**2)))
   write (*, '(A,1P,E12.4)') 'ENORM=', enorm
   write (*, '(A,1P,8E25.17)') 'X8=', x(1:8)

   ! the same operator reached through a user-written Fortran subclass
   ws%inner = p%handle
   call ws%initialize_device(m, n)
   u = p%b
   call ws%lsqr(m, n, damp, .false., u, v, w, x2, se, atol, atol, conlim, itnlim, 0, &
                istop, itn2, anorm, acond, rnorm, arnorm, xnorm)
   write (*, '(A,I2,A,I5,A,I6)') 'USER istop=', istop, ' itn=', itn2, ' calls=', ws%calls
   write (*, '(A,1P,E12.4)') 'USER maxdiff=', maxval(abs(x2 - x))
   if (itn2 /= itn .or. maxval(abs(x2 - x)) /= zero) error stop 'TEST FAILED: user subclass differs'
   if (ws%calls < 2*itn + 1) error stop 'TEST FAILED: callback not used'
   call ws%destroy()
   call p%destroy()
   if (istop /= 3 .or. enorm > merge(1.0e-3_wp, 1.0e-6_wp, single)) error stop 'TEST FAILED'
   write (*, '(A)') 'DEVICE OPERATOR TESTS PASSED'
end program test_device_operator
