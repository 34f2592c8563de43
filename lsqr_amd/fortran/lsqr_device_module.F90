!> LSQR on DEVICE-RESIDENT operators (include/lsqrhip.h "user-supplied DEVICE operator").
!!
!! The reference's abstract class `lsqr_solver` (src/lsqr.f90:16-30) takes a host `aprod`
!! working on host arrays; `lsqr_module` keeps that class as it is.  This module adds its
!! device twin (SURVEY 8f rank 3):
!!
!!   * `lsqr_solver_device`      abstract; the user overrides `aprod_device(me, mode, m, n, x, y,
!!                               stream)` where x (n) and y (m) are DEVICE addresses (type(c_ptr))
!!                               and the override only ENQUEUES  y <- y + A x  (mode 1) or
!!                               x <- x + A' y  (mode 2) on `stream`.  `lsqr`, `acheck`, `xcheck`
!!                               keep the reference's argument lists (src/lsqr.f90:432-435,
!!                               908-909, 1015-1017) and run on the GPU around that callback.
!!   * `lsqr_test_problem_device` the reference's own test operator A = HY*D*HZ with the problem
!!                               generator lstp (test/lsqrtest_module.f90:283-505) as a built-in
!!                               device operator: `create(m, n, nduplc, npower, damp)`.
!!
!! An object that has been initialised must stay where it is (declare it `target`; do not copy
!! it): the library calls back into that very object.
!!
!! Precision (src/lsqr_kinds.F90:16-24 applies to the abstract class too): this file is preprocessed.  Under
!! -DREAL32 (wp = real32) it binds the library's REAL32 operator entry points -- lsqrhip_create_operator_f32,
!! lsqrhip_lstp_create_f32, lsqrhip_solve_f32, lsqrhip_acheck_f32, lsqrhip_xcheck_f32 -- so that x, y of
!! `aprod_device` and every work vector of the iteration are real32 arrays ON THE DEVICE (binary64 registers), as
!! `lsqr_module` does for the EZ class; otherwise the binary64 ones.
#ifdef REAL32
#define CWP c_float
#define NAME_CREATE_OPERATOR 'lsqrhip_create_operator_f32'
#define NAME_LSTP_CREATE 'lsqrhip_lstp_create_f32'
#define NAME_SOLVE 'lsqrhip_solve_f32'
#define NAME_ACHECK 'lsqrhip_acheck_f32'
#define NAME_XCHECK 'lsqrhip_xcheck_f32'
#define NAME_APROD_DEVICE 'lsqrhip_aprod_device_f32'
#else
#define CWP c_double
#define NAME_CREATE_OPERATOR 'lsqrhip_create_operator'
#define NAME_LSTP_CREATE 'lsqrhip_lstp_create'
#define NAME_SOLVE 'lsqrhip_solve'
#define NAME_ACHECK 'lsqrhip_acheck'
#define NAME_XCHECK 'lsqrhip_xcheck'
#define NAME_APROD_DEVICE 'lsqrhip_aprod_device'
#endif
module lsqr_device_module
   use, intrinsic :: iso_c_binding
   use lsqr_kinds
   use lsqr_module, only: lsqr_print_device_log, lsqr_check_status
   implicit none
   private

   type, public :: lsqr_device_handle
      !! what the two public types share: a lsqrhip operator handle and the three solvers' calls
      type(c_ptr) :: handle = c_null_ptr
      integer :: m = 0
      integer :: n = 0
   contains
      procedure, public :: lsqr => lsqr_dev
      procedure, public :: acheck => acheck_dev
      procedure, public :: xcheck => xcheck_dev
      procedure, public :: destroy => destroy_dev
   end type lsqr_device_handle

   type :: dev_box
      class(lsqr_solver_device), pointer :: p => null()
   end type dev_box

   type, abstract, public, extends(lsqr_device_handle) :: lsqr_solver_device
      type(dev_box), pointer, private :: box => null()
   contains
      procedure(aprod_device_func), deferred, public :: aprod_device
      procedure, public :: initialize_device
   end type lsqr_solver_device

   type, public, extends(lsqr_device_handle) :: lsqr_test_problem_device
      real(wp) :: acond = zero      !< condition number reported by lstp
      real(wp) :: rnorm = zero      !< residual function reported by lstp
      real(wp), allocatable :: b(:), xtrue(:)
   contains
      procedure, public :: create => create_test_problem
   end type lsqr_test_problem_device

   abstract interface
      subroutine aprod_device_func(me, mode, m, n, x, y, stream)
         import :: lsqr_solver_device, c_ptr
         implicit none
         class(lsqr_solver_device), intent(inout) :: me
         integer, intent(in) :: mode, m, n
         type(c_ptr), intent(in) :: x        !< device address of x(n)
         type(c_ptr), intent(in) :: y        !< device address of y(m)
         type(c_ptr), intent(in) :: stream   !< hipStream_t to enqueue on
      end subroutine aprod_device_func
   end interface

   interface
      function lsqrhip_create_operator(m, n, aprod, user, h) bind(C, name=NAME_CREATE_OPERATOR) result(rc)
         import :: c_int, c_ptr, c_funptr
         integer(c_int), value :: m, n
         type(c_funptr), value :: aprod
         type(c_ptr), value :: user
         type(c_ptr), intent(out) :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_lstp_create(m, n, nduplc, npower, damp, h, acond, rnorm) &
         bind(C, name=NAME_LSTP_CREATE) result(rc)
         import :: c_int, c_double, c_ptr
         integer(c_int), value :: m, n, nduplc, npower
         real(c_double), value :: damp
         type(c_ptr), intent(out) :: h
         real(c_double), intent(out) :: acond, rnorm
         integer(c_int) :: rc
      end function
      function lsqrhip_lstp_vectors(h, xtrue, b, d, hy, hz, d_b) bind(C, name='lsqrhip_lstp_vectors') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         real(c_double), intent(out) :: xtrue(*), b(*)
         type(c_ptr), value :: d, hy, hz, d_b
         integer(c_int) :: rc
      end function
      function lsqrhip_destroy(h) bind(C, name='lsqrhip_destroy') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_solve(h, b, damp, atol, btol, conlim, itnlim, wantse, want_log, x, se, istop, itn, &
                             anorm, acond, rnorm, arnorm, xnorm) bind(C, name=NAME_SOLVE) result(rc)
         import :: c_int, c_double, c_float, c_ptr
         type(c_ptr), value :: h
         real(CWP), intent(in) :: b(*)
         real(c_double), value :: damp, atol, btol, conlim
         integer(c_int), value :: itnlim, wantse, want_log
         real(CWP), intent(out) :: x(*)
         real(CWP), intent(inout) :: se(*)
         integer(c_int), intent(out) :: istop, itn
         real(c_double), intent(out) :: anorm, acond, rnorm, arnorm, xnorm
         integer(c_int) :: rc
      end function
      function lsqrhip_acheck(h, eps, inform, relerr) bind(C, name=NAME_ACHECK) result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         real(c_double), value :: eps
         integer(c_int), intent(out) :: inform
         real(c_double), intent(out) :: relerr
         integer(c_int) :: rc
      end function
      function lsqrhip_xcheck(h, anorm, damp, eps, b, x, u, v, w, inform, tests) &
         bind(C, name=NAME_XCHECK) result(rc)
         import :: c_int, c_double, c_float, c_ptr
         type(c_ptr), value :: h
         real(c_double), value :: anorm, damp, eps
         real(CWP), intent(in) :: b(*), x(*)
         real(CWP), intent(out) :: u(*), v(*), w(*)
         integer(c_int), intent(out) :: inform
         real(c_double), intent(out) :: tests(3)
         integer(c_int) :: rc
      end function
   end interface

   !> Library calls a Fortran user may want inside `aprod_device` (e.g. to apply another handle's
   !! product on the same stream).
   interface
      function lsqrhip_aprod_device(h, mode, d_x, d_y) bind(C, name=NAME_APROD_DEVICE) result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: h, d_x, d_y
         integer(c_int), value :: mode
         integer(c_int) :: rc
      end function
      function lsqrhip_set_stream(h, stream) bind(C, name='lsqrhip_set_stream') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: h, stream
         integer(c_int) :: rc
      end function
   end interface
   public :: lsqrhip_aprod_device, lsqrhip_set_stream

contains

   !> C entry point handed to lsqrhip_create_operator: forwards to the object's aprod_device.
   function trampoline(user, mode, m, n, d_x, d_y, stream) bind(C) result(rc)
      type(c_ptr), value :: user, d_x, d_y, stream
      integer(c_int), value :: mode, m, n
      integer(c_int) :: rc
      type(dev_box), pointer :: box
      call c_f_pointer(user, box)
      call box%p%aprod_device(int(mode), int(m), int(n), d_x, d_y, stream)
      rc = 0_c_int
   end function trampoline

   !> Register the object's operator with the library (the constructor of the device class).
   subroutine initialize_device(me, m, n)
      class(lsqr_solver_device), intent(inout), target :: me
      integer, intent(in) :: m, n
      call me%destroy()
      allocate (me%box)
      me%box%p => me
      me%m = m
      me%n = n
      call lsqr_check_status(lsqrhip_create_operator(int(m, c_int), int(n, c_int), c_funloc(trampoline), &
                                                      c_loc(me%box), me%handle))
   end subroutine initialize_device

   subroutine destroy_dev(me)
      class(lsqr_device_handle), intent(inout) :: me
      integer(c_int) :: rc
      if (c_associated(me%handle)) rc = lsqrhip_destroy(me%handle)
      me%handle = c_null_ptr
      select type (me)
      class is (lsqr_solver_device)
         if (associated(me%box)) deallocate (me%box)
      end select
   end subroutine destroy_dev

   !> The reference's test problem P(m, n, nduplc, npower, damp) (test/lsqrtest_module.f90:422-505).
   subroutine create_test_problem(me, m, n, nduplc, npower, damp)
      class(lsqr_test_problem_device), intent(inout) :: me
      integer, intent(in) :: m, n, nduplc, npower
      real(wp), intent(in) :: damp
      real(c_double) :: ac, rn
      real(c_double), allocatable :: bl(:), xl(:)
      call me%destroy()
      me%m = m
      me%n = n
      call lsqr_check_status(lsqrhip_lstp_create(int(m, c_int), int(n, c_int), int(nduplc, c_int), &
                                                  int(npower, c_int), real(damp, c_double), me%handle, ac, rn))
      me%acond = real(ac, wp)
      me%rnorm = real(rn, wp)
      if (allocated(me%b)) deallocate (me%b)
      if (allocated(me%xtrue)) deallocate (me%xtrue)
      allocate (me%b(m), me%xtrue(n), bl(m), xl(n))
      call lsqr_check_status(lsqrhip_lstp_vectors(me%handle, xl, bl, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr))
      me%b = real(bl, wp)
      me%xtrue = real(xl, wp)
   end subroutine create_test_problem

   !> LSQR (src/lsqr.f90:432-882) on the device operator.  Argument list of the reference:
   !! `u` holds b on entry; `v`, `w` are accepted for compatibility (the work vectors live in
   !! HBM and are not copied back); `se` is referenced only when `wantse`.
   subroutine lsqr_dev(me, m, n, damp, wantse, u, v, w, x, se, atol, btol, conlim, itnlim, nout, &
                       istop, itn, anorm, acond, rnorm, arnorm, xnorm)
      class(lsqr_device_handle), intent(inout) :: me
      integer, intent(in) :: m, n, itnlim, nout
      real(wp), intent(in) :: damp, atol, btol, conlim
      logical, intent(in) :: wantse
      real(wp), intent(inout) :: u(m), v(n), w(n)
      real(wp), intent(out) :: x(n)
      real(wp), intent(inout) :: se(*)
      integer, intent(out) :: istop, itn
      real(wp), intent(out) :: anorm, acond, rnorm, arnorm, xnorm
      integer(c_int) :: istop_, itn_
      real(c_double) :: sc(5)
      real(CWP), allocatable :: bl(:), xl(:), sel(:)
      if (.not. c_associated(me%handle) .or. m /= me%m .or. n /= me%n) call lsqr_check_status(4_c_int)
      allocate (bl(max(m, 1)), xl(max(n, 1)), sel(max(n, 1)))
      bl(1:m) = u(1:m)
      call lsqr_check_status(lsqrhip_solve(me%handle, bl, real(damp, c_double), real(atol, c_double), &
                                           real(btol, c_double), real(conlim, c_double), int(itnlim, c_int), &
                                           merge(1_c_int, 0_c_int, wantse), merge(1_c_int, 0_c_int, nout /= 0), xl, sel, &
                                           istop_, itn_, sc(1), sc(2), sc(3), sc(4), sc(5)))
      x(1:n) = real(xl(1:n), wp)
      if (wantse) se(1:n) = real(sel(1:n), wp)
      anorm = real(sc(1), wp)
      acond = real(sc(2), wp)
      rnorm = real(sc(3), wp)
      arnorm = real(sc(4), wp)
      xnorm = real(sc(5), wp)
      istop = istop_
      itn = itn_
      if (nout /= 0) call lsqr_print_device_log(me%handle, nout, m, n, damp, wantse, atol, btol, conlim, itnlim, &
                                                istop, itn, anorm, acond, rnorm, arnorm, xnorm)
      if (.false.) v(1) = w(1)   ! v, w: compatibility arguments
   end subroutine lsqr_dev

   !> acheck (src/lsqr.f90:908-994) on the device operator; v, w, x, y are compatibility arguments.
   subroutine acheck_dev(me, m, n, nout, eps, v, w, x, y, inform)
      class(lsqr_device_handle), intent(inout) :: me
      integer, intent(in) :: m, n, nout
      real(wp), intent(in) :: eps
      real(wp), intent(inout) :: v(n), w(m), x(n), y(m)
      integer, intent(out) :: inform
      integer(c_int) :: inf
      real(c_double) :: err
      if (.not. c_associated(me%handle) .or. m /= me%m .or. n /= me%n) call lsqr_check_status(4_c_int)
      call lsqr_check_status(lsqrhip_acheck(me%handle, real(eps, c_double), inf, err))
      inform = inf
      if (nout /= 0) then
         write (nout, '(//A)') ' Enter acheck.     Test of aprod for LSQR and CRAIG'
         if (inform == 0) then
            write (nout, '(1P,A,1X,E10.1)') ' aprod seems OK.   Relative error =', err
         else
            write (nout, '(1P,A,1X,E10.1)') ' aprod seems incorrect.   Relative error =', err
         end if
      end if
      if (.false.) v(1) = w(1) + x(1) + y(1)
   end subroutine acheck_dev

   !> xcheck (src/lsqr.f90:1015-1154) on the device operator: u = b - A x, v = A'u, w = v - damp^2 x.
   subroutine xcheck_dev(me, m, n, nout, anorm, damp, eps, b, u, v, w, x, inform, test1, test2, test3)
      class(lsqr_device_handle), intent(inout) :: me
      integer, intent(in) :: m, n, nout
      real(wp), intent(in) :: anorm, damp, eps
      real(wp), intent(in) :: b(m), x(n)
      real(wp), intent(out) :: u(m), v(n), w(n)
      integer, intent(out) :: inform
      real(wp), intent(out) :: test1, test2, test3
      integer(c_int) :: inf
      real(c_double) :: tests(3)
      real(CWP), allocatable :: bl(:), xl(:), ul(:), vl(:), wl(:)
      if (.not. c_associated(me%handle) .or. m /= me%m .or. n /= me%n) call lsqr_check_status(4_c_int)
      allocate (bl(max(m, 1)), xl(max(n, 1)), ul(max(m, 1)), vl(max(n, 1)), wl(max(n, 1)))
      bl(1:m) = b
      xl(1:n) = x
      call lsqr_check_status(lsqrhip_xcheck(me%handle, real(anorm, c_double), real(damp, c_double), &
                                            real(eps, c_double), bl, xl, ul, vl, wl, inf, tests))
      u = real(ul(1:m), wp)
      v = real(vl(1:n), wp)
      w = real(wl(1:n), wp)
      inform = inf
      test1 = real(tests(1), wp)
      test2 = real(tests(2), wp)
      test3 = real(tests(3), wp)
      if (nout /= 0) then
         write (nout, '(//A)') ' Enter xcheck.     Does x solve Ax = b, etc?'
         write (nout, '(/A,I2)') '    inform          =', inform
         write (nout, '(1P,A,E10.3)') '    tol             =', sqrt(eps)
         write (nout, '(1P,A,E10.3,A)') '    test1           =', test1, ' (Ax = b)'
         write (nout, '(1P,A,E10.3,A)') '    test2           =', test2, ' (least-squares)'
         write (nout, '(1P,A,E10.3,A)') '    test3           =', test3, ' (damped least-squares)'
      end if
   end subroutine xcheck_dev

end module lsqr_device_module
