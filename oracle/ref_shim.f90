!  oracle/ref_shim.f90 -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
!
!  bind(C) entry points around the UNMODIFIED reference modules (lsqr_kinds,
!  lsqpblas_module, lsqr_module), which oracle/Makefile compiles in place from
!  /root/reference/src into oracle/_ref/.  This file is our own code: it only
!  `use`s the reference's public API (lsqr_solver_ez%initialize/solve/aprod/
!  acheck/xcheck and the four BLAS-1 routines) so that Python (ctypes) can call
!  the real reference to (a) pin oracle/lsqr_oracle.c, (b) generate the fixtures
!  in tests/golden/, (c) serve as bench.py's cpu_baseline ("kind": "reference").

module ref_shim
   use iso_c_binding
   use lsqr_kinds, only: wp, zero
   use lsqpblas_module, only: dcopy, ddot, dnrm2, dscal
   use lsqr_module, only: lsqr_solver_ez
   implicit none
   private
contains

   !> lsqr_solver_ez%initialize + %solve   (src/lsqr.f90:91-127, 207-259)
   subroutine ref_lsqr_ez(m, n, nnz, irow, icol, a, b, damp, atol, btol, conlim, itnlim, &
                          wantse, x, se, istop, itn, anorm, acond, rnorm, arnorm, xnorm, &
                          logpath, loglen) bind(C, name='ref_lsqr_ez')
      integer(c_int), value :: m, n, nnz, itnlim, wantse, loglen
      integer(c_int), intent(in) :: irow(nnz), icol(nnz)
      real(c_double), intent(in) :: a(nnz), b(m)
      real(c_double), value :: damp, atol, btol, conlim
      real(c_double), intent(out) :: x(n)
      real(c_double), intent(inout) :: se(*)
      integer(c_int), intent(out) :: istop, itn
      real(c_double), intent(out) :: anorm, acond, rnorm, arnorm, xnorm
      character(kind=c_char), intent(in) :: logpath(*)

      type(lsqr_solver_ez) :: solver
      integer :: nout, i
      character(len=:), allocatable :: path

      nout = 0
      if (loglen > 0) then
         allocate (character(len=loglen) :: path)
         do i = 1, loglen
            path(i:i) = logpath(i)
         end do
         open (newunit=nout, file=path, status='replace', action='write')
      end if

      ! rnorm is left unassigned by the reference when the loop is skipped
      ! (src/lsqr.f90:646-653); pre-set the locals it would leave undefined.
      anorm = zero; acond = zero; rnorm = zero; arnorm = zero; xnorm = zero; itn = 0

      call solver%initialize(m, n, a, irow, icol, atol=atol, btol=btol, conlim=conlim, &
                             itnlim=itnlim, nout=nout)
      if (wantse /= 0) then
         call solver%solve(b, damp, x, istop, se=se(1:n), itn=itn, anorm=anorm, acond=acond, &
                           rnorm=rnorm, arnorm=arnorm, xnorm=xnorm)
      else
         call solver%solve(b, damp, x, istop, itn=itn, anorm=anorm, acond=acond, &
                           rnorm=rnorm, arnorm=arnorm, xnorm=xnorm)
      end if
      if (nout /= 0) close (nout)
   end subroutine ref_lsqr_ez

   !> lsqr_solver_ez%aprod   (src/lsqr.f90:134-200)
   subroutine ref_aprod(mode, m, n, nnz, irow, icol, a, x, y) bind(C, name='ref_aprod')
      integer(c_int), value :: mode, m, n, nnz
      integer(c_int), intent(in) :: irow(nnz), icol(nnz)
      real(c_double), intent(in) :: a(nnz)
      real(c_double), intent(inout) :: x(n), y(m)
      type(lsqr_solver_ez) :: solver
      call solver%initialize(m, n, a, irow, icol)
      call solver%aprod(mode, m, n, x, y)
   end subroutine ref_aprod

   !> lsqr_solver%acheck on the EZ operator   (src/lsqr.f90:908-994)
   subroutine ref_acheck(m, n, nnz, irow, icol, a, eps, inform) bind(C, name='ref_acheck')
      integer(c_int), value :: m, n, nnz
      integer(c_int), intent(in) :: irow(nnz), icol(nnz)
      real(c_double), intent(in) :: a(nnz)
      real(c_double), value :: eps
      integer(c_int), intent(out) :: inform
      type(lsqr_solver_ez) :: solver
      real(wp), allocatable :: v(:), w(:), x(:), y(:)
      allocate (v(n), w(m), x(n), y(m))
      call solver%initialize(m, n, a, irow, icol)
      call solver%acheck(m, n, 0, eps, v, w, x, y, inform)
   end subroutine ref_acheck

   !> lsqr_solver%xcheck on the EZ operator   (src/lsqr.f90:1015-1154)
   subroutine ref_xcheck(m, n, nnz, irow, icol, a, anorm, damp, eps, b, x, u, v, w, inform, tests) &
      bind(C, name='ref_xcheck')
      integer(c_int), value :: m, n, nnz
      integer(c_int), intent(in) :: irow(nnz), icol(nnz)
      real(c_double), intent(in) :: a(nnz), b(m), x(n)
      real(c_double), value :: anorm, damp, eps
      real(c_double), intent(out) :: u(m), v(n), w(n), tests(3)
      integer(c_int), intent(out) :: inform
      type(lsqr_solver_ez) :: solver
      call solver%initialize(m, n, a, irow, icol)
      call solver%xcheck(m, n, 0, anorm, damp, eps, b, u, v, w, x, inform, tests(1), tests(2), tests(3))
   end subroutine ref_xcheck

   !> BLAS-1   (src/lsqrblas.f90:25-201)
   function ref_dnrm2(n, x, incx) result(r) bind(C, name='ref_dnrm2')
      integer(c_int), value :: n, incx
      real(c_double) :: x(*)
      real(c_double) :: r
      r = dnrm2(n, x, incx)
   end function ref_dnrm2

   function ref_ddot(n, x, incx, y, incy) result(r) bind(C, name='ref_ddot')
      integer(c_int), value :: n, incx, incy
      real(c_double) :: x(*), y(*)
      real(c_double) :: r
      r = ddot(n, x, incx, y, incy)
   end function ref_ddot

   subroutine ref_dscal(n, da, x, incx) bind(C, name='ref_dscal')
      integer(c_int), value :: n, incx
      real(c_double), value :: da
      real(c_double) :: x(*)
      real(c_double) :: da_
      da_ = da
      call dscal(n, da_, x, incx)
   end subroutine ref_dscal

   subroutine ref_dcopy(n, x, incx, y, incy) bind(C, name='ref_dcopy')
      integer(c_int), value :: n, incx, incy
      real(c_double) :: x(*), y(*)
      call dcopy(n, x, incx, y, incy)
   end subroutine ref_dcopy

end module ref_shim
