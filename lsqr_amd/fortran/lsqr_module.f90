!> LSQR host layer: the reference's two public types over the MI355X C-ABI.
!!
!! Source-compatible with the reference's `lsqr_module` (src/lsqr.f90:14-82):
!!
!!   * `lsqr_solver`     abstract; the user supplies `aprod`.  `lsqr`, `acheck`, `xcheck`
!!                       run on the host around that callback (the callback is host code).
!!   * `lsqr_solver_ez`  the matrix is given as COO triplets.  `initialize` hands them to
!!                       liblsqrhip.so (include/lsqrhip.h), which builds CSR(A) and CSR(A') in
!!                       HBM; `solve` runs the whole Golub-Kahan iteration on the GPU; `aprod`
!!                       is one fused SpMV launch.  No CPU fallback: without a device these
!!                       `error stop`.
!!
!! Argument lists, optional arguments, `error stop` strings and the iteration-log text
!! follow the reference; everything else (structure, kernels, numerics of the norms) is new.
module lsqr_module
   use, intrinsic :: iso_c_binding
   use lsqr_kinds
   use lsqpblas_module
   implicit none
   private

   integer, parameter :: LOG_STRIDE = 14   !< LSQRHIP_LOG_STRIDE

   public :: lsqr_print_device_log, lsqr_check_status

   type, abstract, public :: lsqr_solver
      !! Solver driven by a user-written operator (reference src/lsqr.f90:16-30).
   contains
      procedure(aprod_func), deferred, public :: aprod
      procedure, public :: lsqr
      procedure, public :: acheck
      procedure, public :: xcheck
   end type lsqr_solver

   type, public, extends(lsqr_solver) :: lsqr_solver_ez
      !! COO matrix living on the GPU (reference src/lsqr.f90:32-65).
      private
      integer  :: m = 0
      integer  :: n = 0
      integer  :: num_nonzero_elements = 0
      real(wp) :: atol = zero
      real(wp) :: btol = zero
      real(wp) :: conlim = zero
      integer  :: itnlim = 100
      integer  :: nout = 0
      type(c_ptr) :: handle = c_null_ptr   !< lsqrhip_handle_t (reference-counted, see assignment)
      logical  :: io32 = .false.           !< REAL32 build, one GPU: a handle of lsqrhip_create_f32 (real32 on the device)
      logical  :: sharded = .false.        !< rows cut over several GPUs (`initialize(..., ngpu=)`)
   contains
      procedure, public :: initialize => initialize_ez
      procedure, public :: solve => solve_ez
      procedure, public :: aprod => aprod_ez
      procedure, public :: lsqr => lsqr_ez       !< the inherited entry points, when called on THIS type, run on the
      procedure, public :: acheck => acheck_ez   !< device too (round 5): no aprod round trip over PCIe per product
      procedure, public :: xcheck => xcheck_ez
      procedure, public :: destroy => destroy_ez
      procedure, private :: copy_ez
      generic, public :: assignment(=) => copy_ez
      final :: finalize_ez
   end type lsqr_solver_ez

   abstract interface
      subroutine aprod_func(me, mode, m, n, x, y)
         !! mode 1: y = y + A*x (x unchanged); mode 2: x = x + A'*y (y unchanged).
         import :: wp, lsqr_solver
         implicit none
         class(lsqr_solver), intent(inout) :: me
         integer, intent(in) :: mode, m, n
         real(wp), dimension(:), intent(inout) :: x
         real(wp), dimension(:), intent(inout) :: y
      end subroutine aprod_func
   end interface

   ! ---- C-ABI (include/lsqrhip.h) ---------------------------------------------------------
   interface
      function lsqrhip_create(m, n, nnz, irow, icol, a, h) bind(C, name='lsqrhip_create') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         integer(c_int), value :: m, n
         integer(c_int64_t), value :: nnz
         integer(c_int), intent(in) :: irow(*), icol(*)
         real(c_double), intent(in) :: a(*)
         type(c_ptr), intent(out) :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_create_sharded(m, n, nnz, irow, icol, a, ngpu, h) bind(C, name='lsqrhip_create_sharded') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         integer(c_int), value :: m, n, ngpu
         integer(c_int64_t), value :: nnz
         integer(c_int), intent(in) :: irow(*), icol(*)
         real(c_double), intent(in) :: a(*)
         type(c_ptr), intent(out) :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_create_sharded_f32(m, n, nnz, irow, icol, a, ngpu, h) bind(C, name='lsqrhip_create_sharded_f32') result(rc)
         import :: c_int, c_int64_t, c_float, c_ptr
         integer(c_int), value :: m, n, ngpu
         integer(c_int64_t), value :: nnz
         integer(c_int), intent(in) :: irow(*), icol(*)
         real(c_float), intent(in) :: a(*)
         type(c_ptr), intent(out) :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_destroy(h) bind(C, name='lsqrhip_destroy') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_retain(h) bind(C, name='lsqrhip_retain') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_solve(h, b, damp, atol, btol, conlim, itnlim, wantse, want_log, x, se, istop, itn, &
                             anorm, acond, rnorm, arnorm, xnorm) bind(C, name='lsqrhip_solve') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         real(c_double), intent(in) :: b(*)
         real(c_double), value :: damp, atol, btol, conlim
         integer(c_int), value :: itnlim, wantse, want_log
         real(c_double), intent(out) :: x(*)
         real(c_double), intent(inout) :: se(*)
         integer(c_int), intent(out) :: istop, itn
         real(c_double), intent(out) :: anorm, acond, rnorm, arnorm, xnorm
         integer(c_int) :: rc
      end function
      function lsqrhip_aprod(h, mode, x, y) bind(C, name='lsqrhip_aprod') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         integer(c_int), value :: mode
         real(c_double), intent(inout) :: x(*), y(*)
         integer(c_int) :: rc
      end function
      ! the REAL32 build's entry points (include/lsqrhip.h, "REAL32"): real32 arrays, real32 on the device
      function lsqrhip_create_f32(m, n, nnz, irow, icol, a, h) bind(C, name='lsqrhip_create_f32') result(rc)
         import :: c_int, c_int64_t, c_float, c_ptr
         integer(c_int), value :: m, n
         integer(c_int64_t), value :: nnz
         integer(c_int), intent(in) :: irow(*), icol(*)
         real(c_float), intent(in) :: a(*)
         type(c_ptr), intent(out) :: h
         integer(c_int) :: rc
      end function
      function lsqrhip_solve_f32(h, b, damp, atol, btol, conlim, itnlim, wantse, want_log, x, se, istop, itn, &
                                 anorm, acond, rnorm, arnorm, xnorm) bind(C, name='lsqrhip_solve_f32') result(rc)
         import :: c_int, c_double, c_float, c_ptr
         type(c_ptr), value :: h
         real(c_float), intent(in) :: b(*)
         real(c_double), value :: damp, atol, btol, conlim
         integer(c_int), value :: itnlim, wantse, want_log
         real(c_float), intent(out) :: x(*)
         real(c_float), intent(inout) :: se(*)
         integer(c_int), intent(out) :: istop, itn
         real(c_double), intent(out) :: anorm, acond, rnorm, arnorm, xnorm
         integer(c_int) :: rc
      end function
      function lsqrhip_aprod_f32(h, mode, x, y) bind(C, name='lsqrhip_aprod_f32') result(rc)
         import :: c_int, c_float, c_ptr
         type(c_ptr), value :: h
         integer(c_int), value :: mode
         real(c_float), intent(inout) :: x(*), y(*)
         integer(c_int) :: rc
      end function
      function lsqrhip_acheck(h, eps, inform, relerr) bind(C, name='lsqrhip_acheck') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         real(c_double), value :: eps
         integer(c_int), intent(out) :: inform
         real(c_double), intent(out) :: relerr
         integer(c_int) :: rc
      end function
      function lsqrhip_acheck_f32(h, eps, inform, relerr) bind(C, name='lsqrhip_acheck_f32') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         real(c_double), value :: eps
         integer(c_int), intent(out) :: inform
         real(c_double), intent(out) :: relerr
         integer(c_int) :: rc
      end function
      function lsqrhip_xcheck(h, anorm, damp, eps, b, x, u, v, w, inform, tests) bind(C, name='lsqrhip_xcheck') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         real(c_double), value :: anorm, damp, eps
         real(c_double), intent(in) :: b(*), x(*)
         real(c_double), intent(out) :: u(*), v(*), w(*)
         integer(c_int), intent(out) :: inform
         real(c_double), intent(out) :: tests(3)
         integer(c_int) :: rc
      end function
      function lsqrhip_xcheck_f32(h, anorm, damp, eps, b, x, u, v, w, inform, tests) bind(C, name='lsqrhip_xcheck_f32') result(rc)
         import :: c_int, c_double, c_float, c_ptr
         type(c_ptr), value :: h
         real(c_double), value :: anorm, damp, eps
         real(c_float), intent(in) :: b(*), x(*)
         real(c_float), intent(out) :: u(*), v(*), w(*)
         integer(c_int), intent(out) :: inform
         real(c_double), intent(out) :: tests(3)
         integer(c_int) :: rc
      end function
      function lsqrhip_log_count(h) bind(C, name='lsqrhip_log_count') result(k)
         import :: c_int, c_ptr
         type(c_ptr), value :: h
         integer(c_int) :: k
      end function
      function lsqrhip_log_fetch(h, first, count, records) bind(C, name='lsqrhip_log_fetch') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         integer(c_int), value :: first, count
         real(c_double), intent(out) :: records(*)
         integer(c_int) :: rc
      end function
      function lsqrhip_log_extras(h, ex) bind(C, name='lsqrhip_log_extras') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: h
         real(c_double), intent(out) :: ex(*)
         integer(c_int) :: rc
      end function
      function lsqrhip_last_error() bind(C, name='lsqrhip_last_error') result(p)
         import :: c_ptr
         type(c_ptr) :: p
      end function
      function c_strlen(s) bind(C, name='strlen') result(k)
         import :: c_ptr, c_size_t
         type(c_ptr), value :: s
         integer(c_size_t) :: k
      end function
   end interface

   character(len=*), parameter :: enter_tag = ' Enter LSQR.  '
   character(len=*), parameter :: exit_tag = ' Exit  LSQR.  '
   character(len=*), parameter :: iter_fmt = '(1P, I6, 2E17.9, 4E10.2, E9.1, 3E8.1)'
   character(len=53), parameter :: stop_msg(0:5) = [ &
      'The exact solution is x = 0                          ', &
      'A solution to Ax = b was found, given atol, btol     ', &
      'A least-squares solution was found, given atol       ', &
      'A damped least-squares solution was found, given atol', &
      'Cond(Abar) seems to be too large, given conlim       ', &
      'The iteration limit was reached                      ']

contains

   ! =========================================================================================
   !  lsqr_solver_ez : GPU path
   ! =========================================================================================

   !> Map a C-ABI status to the reference's `error stop` (src/lsqr.f90:109-111, 152, 197).
   subroutine check(rc)
      integer(c_int), intent(in) :: rc
      type(c_ptr) :: p
      character(kind=c_char), pointer :: s(:)
      integer :: k, i
      select case (rc)
      case (0)
         return
      case (1)
         error stop 'invalid a,icol,irow sizes in initialize_ez'
      case (2)
         error stop 'invalid irow or m in initialize_ez'
      case (3)
         error stop 'invalid icol or n in initialize_ez'
      case (4)
         error stop 'lsqr_solver_ez class not properly initialized'
      case (5)
         error stop 'invalid mode in aprod_ez'
      case default
         p = lsqrhip_last_error()
         if (c_associated(p)) then
            k = int(c_strlen(p))
            call c_f_pointer(p, s, [k])
            write (*, '(A)', advance='no') ' lsqrhip: '
            do i = 1, k
               write (*, '(A)', advance='no') s(i)
            end do
            write (*, *)
         end if
         if (rc == 10) error stop 'lsqr_solver_ez: no usable MI355X (gfx950) device; there is no CPU fallback'
         error stop 'lsqr_solver_ez: HIP runtime failure'
      end select
   end subroutine check

   !> `check` for sibling modules (lsqr_device_module).
   subroutine lsqr_check_status(rc)
      integer(c_int), intent(in) :: rc
      call check(rc)
   end subroutine lsqr_check_status

   !> Constructor (replaces src/lsqr.f90:91-127).  `me` is intent(out): any matrix the object
   !! held before is released (finalisation) and every option returns to its default.
   !!
   !! `ngpu` (optional, last, not in the reference): the number of GPUs of this node to shard the
   !! rows over (contiguous row blocks balanced by nonzeros, RCCL over xGMI inside the library).
   !! Absent: one GPU.  More GPUs than the node has: `error stop`.
   subroutine initialize_ez(me, m, n, a, irow, icol, atol, btol, conlim, itnlim, nout, ngpu)
      class(lsqr_solver_ez), intent(out) :: me
      integer, intent(in) :: m, n
      integer, dimension(:), intent(in) :: irow, icol
      real(wp), dimension(:), intent(in) :: a
      real(wp), intent(in), optional :: atol, btol, conlim
      integer, intent(in), optional :: itnlim, nout
      integer, intent(in), optional :: ngpu
      integer(c_int), allocatable :: ir(:), ic(:)
      real(c_double), allocatable :: av(:)
      real(c_float), allocatable :: af(:)
      integer(c_int64_t) :: nz

      if (wp /= c_float .and. wp /= c_double) &
         error stop 'lsqr_solver_ez: the REAL128 build has no device path (the GPU computes in binary64); use lsqr_solver with a host aprod'
      if (size(a) /= size(irow) .or. size(a) /= size(icol)) call check(1_c_int)
      ir = irow            ! contiguous copies for the C side
      ic = icol
      nz = int(size(a), c_int64_t)
      if (size(ir) == 0) then
         deallocate (ir, ic)
         allocate (ir(1), ic(1))
      end if
      if (wp == c_float) then
         ! the REAL32 build: real32 storage on the device(s) too, and on the links of a sharded solve
         ! (binary64 in registers only)
         allocate (af(max(size(a), 1)))
         af(1:size(a)) = real(a, c_float)
         if (present(ngpu)) then
            call check(lsqrhip_create_sharded_f32(int(m, c_int), int(n, c_int), nz, ir, ic, af, int(ngpu, c_int), me%handle))
            me%sharded = .true.
         else
            call check(lsqrhip_create_f32(int(m, c_int), int(n, c_int), nz, ir, ic, af, me%handle))
         end if
         me%io32 = .true.
      else
         allocate (av(max(size(a), 1)))
         av(1:size(a)) = real(a, c_double)
         if (present(ngpu)) then
            call check(lsqrhip_create_sharded(int(m, c_int), int(n, c_int), nz, ir, ic, av, int(ngpu, c_int), me%handle))
            me%sharded = .true.
         else
            call check(lsqrhip_create(int(m, c_int), int(n, c_int), nz, ir, ic, av, me%handle))
         end if
      end if
      me%m = m
      me%n = n
      me%num_nonzero_elements = size(a)
      if (present(atol)) me%atol = atol
      if (present(btol)) me%btol = btol
      if (present(conlim)) me%conlim = conlim
      if (present(itnlim)) me%itnlim = itnlim
      if (present(nout)) me%nout = nout
   end subroutine initialize_ez

   !> Release the device matrix now (optional; finalisation does it too).
   subroutine destroy_ez(me)
      class(lsqr_solver_ez), intent(inout) :: me
      integer(c_int) :: rc
      if (c_associated(me%handle)) rc = lsqrhip_destroy(me%handle)
      me%handle = c_null_ptr
   end subroutine destroy_ez

   subroutine finalize_ez(me)
      type(lsqr_solver_ez), intent(inout) :: me
      call destroy_ez(me)
   end subroutine finalize_ez

   !> Intrinsic assignment would alias the device handle; share it by reference count instead.
   subroutine copy_ez(lhs, rhs)
      class(lsqr_solver_ez), intent(inout) :: lhs
      class(lsqr_solver_ez), intent(in) :: rhs
      integer(c_int) :: rc
      if (c_associated(lhs%handle, rhs%handle)) return
      call destroy_ez(lhs)
      lhs%m = rhs%m; lhs%n = rhs%n; lhs%num_nonzero_elements = rhs%num_nonzero_elements
      lhs%atol = rhs%atol; lhs%btol = rhs%btol; lhs%conlim = rhs%conlim
      lhs%itnlim = rhs%itnlim; lhs%nout = rhs%nout
      lhs%sharded = rhs%sharded
      lhs%io32 = rhs%io32
      lhs%handle = rhs%handle
      if (c_associated(lhs%handle)) rc = lsqrhip_retain(lhs%handle)
   end subroutine copy_ez

   !> y = y + A x  or  x = x + A' y on the GPU (replaces src/lsqr.f90:134-200).
   subroutine aprod_ez(me, mode, m, n, x, y)
      class(lsqr_solver_ez), intent(inout) :: me
      integer, intent(in) :: mode, m, n
      real(wp), dimension(:), intent(inout) :: x   !! [n]
      real(wp), dimension(:), intent(inout) :: y   !! [m]
      real(c_double), allocatable :: xl(:), yl(:)
      real(c_float), allocatable :: xf(:), yf(:)
      if (m /= me%m .or. n /= me%n .or. .not. c_associated(me%handle)) call check(4_c_int)
      if (me%io32) then
         allocate (xf(max(n, 1)), yf(max(m, 1)))
         xf(1:n) = real(x(1:n), c_float)
         yf(1:m) = real(y(1:m), c_float)
         call check(lsqrhip_aprod_f32(me%handle, int(mode, c_int), xf, yf))
         if (mode == 1) then
            y(1:m) = real(yf(1:m), wp)
         else
            x(1:n) = real(xf(1:n), wp)
         end if
         return
      end if
      allocate (xl(max(n, 1)), yl(max(m, 1)))
      xl(1:n) = x(1:n)
      yl(1:m) = y(1:m)
      call check(lsqrhip_aprod(me%handle, int(mode, c_int), xl, yl))
      if (mode == 1) then
         y(1:m) = real(yl(1:m), wp)
      else
         x(1:n) = real(xl(1:n), wp)
      end if
   end subroutine aprod_ez

   !> Solve on the GPU (replaces src/lsqr.f90:207-259 and, for this type, the call of :432-882).
   subroutine solve_ez(me, b, damp, x, istop, se, itn, anorm, acond, rnorm, arnorm, xnorm)
      class(lsqr_solver_ez), intent(inout) :: me
      real(wp), dimension(me%m), intent(in) :: b
      real(wp), intent(in) :: damp
      real(wp), dimension(me%n), intent(out) :: x
      integer, intent(out) :: istop
      real(wp), dimension(me%n), intent(out), optional :: se
      integer, intent(out), optional :: itn
      real(wp), intent(out), optional :: anorm, acond, rnorm, arnorm, xnorm

      real(c_double), allocatable :: xl(:), sel(:), bl(:)
      real(c_float), allocatable :: xf(:), sef(:), bf(:)
      integer(c_int) :: istop_, itn_, wantse, want_log
      real(c_double) :: anorm_, acond_, rnorm_, arnorm_, xnorm_

      if (.not. c_associated(me%handle)) call check(4_c_int)
      wantse = merge(1_c_int, 0_c_int, present(se))
      want_log = merge(1_c_int, 0_c_int, me%nout /= 0)   ! (sharded: rank 0 keeps the records, the scalars are replicated)
      if (me%io32) then
         allocate (xf(max(me%n, 1)), sef(max(me%n, 1)), bf(max(me%m, 1)))
         bf(1:me%m) = real(b, c_float)
         call check(lsqrhip_solve_f32(me%handle, bf, real(damp, c_double), real(me%atol, c_double), &
                                      real(me%btol, c_double), real(me%conlim, c_double), int(me%itnlim, c_int), wantse, &
                                      want_log, xf, sef, istop_, itn_, anorm_, acond_, rnorm_, arnorm_, xnorm_))
         x = real(xf(1:me%n), wp)
         if (present(se)) se = real(sef(1:me%n), wp)
      else
         allocate (xl(max(me%n, 1)), sel(max(me%n, 1)), bl(max(me%m, 1)))
         bl(1:me%m) = b
         call check(lsqrhip_solve(me%handle, bl, real(damp, c_double), real(me%atol, c_double), &
                                  real(me%btol, c_double), real(me%conlim, c_double), int(me%itnlim, c_int), wantse, &
                                  want_log, xl, sel, istop_, itn_, anorm_, acond_, rnorm_, arnorm_, xnorm_))
         x = real(xl(1:me%n), wp)
         if (present(se)) se = real(sel(1:me%n), wp)
      end if
      istop = istop_
      if (present(itn)) itn = itn_
      if (present(anorm)) anorm = anorm_
      if (present(acond)) acond = acond_
      if (present(rnorm)) rnorm = rnorm_
      if (present(arnorm)) arnorm = arnorm_
      if (present(xnorm)) xnorm = xnorm_
      if (me%nout /= 0) &
         call print_device_log(me, damp, present(se), int(istop_), int(itn_), real(anorm_, wp), &
                                              real(acond_, wp), real(rnorm_, wp), real(arnorm_, wp), real(xnorm_, wp))
   end subroutine solve_ez

   !> Write the iteration log of a GPU solve from the device records, with the reference's
   !! selective-print rule and format strings (src/lsqr.f90:589-595, 655-671, 813-837, 872-880).
   subroutine print_device_log(me, damp, wantse, istop, itn, anorm, acond, rnorm, arnorm, xnorm)
      class(lsqr_solver_ez), intent(in) :: me
      real(wp), intent(in) :: damp, anorm, acond, rnorm, arnorm, xnorm
      logical, intent(in) :: wantse
      integer, intent(in) :: istop, itn
      call lsqr_print_device_log(me%handle, me%nout, me%m, me%n, damp, wantse, me%atol, me%btol, me%conlim, &
                                 me%itnlim, istop, itn, anorm, acond, rnorm, arnorm, xnorm)
   end subroutine print_device_log

   !> The same for any lsqrhip handle (used by lsqr_device_module for operator handles).
   subroutine lsqr_print_device_log(handle, nout, m, n, damp, wantse, atol, btol, conlim, itnlim, &
                                    istop, itn, anorm, acond, rnorm, arnorm, xnorm)
      type(c_ptr), intent(in) :: handle
      integer, intent(in) :: nout, m, n, itnlim, istop, itn
      real(wp), intent(in) :: damp, atol, btol, conlim, anorm, acond, rnorm, arnorm, xnorm
      logical, intent(in) :: wantse
      real(c_double), allocatable :: rec(:, :)
      real(c_double) :: ex(6)
      real(wp) :: ctol, test3
      integer :: k, nrec, it, ist
      logical :: show

      call log_header(nout, m, n, damp, wantse, atol, btol, conlim, itnlim)
      call check(lsqrhip_log_extras(handle, ex))
      nrec = lsqrhip_log_count(handle)
      ctol = zero
      if (conlim > zero) ctol = one/conlim
      if (itn > 0 .or. istop /= 0) then
         call log_titles(nout, damp > zero, real(ex(5), wp), real(ex(6), wp))
         if (nrec > 0) then
            allocate (rec(LOG_STRIDE, nrec))
            call check(lsqrhip_log_fetch(handle, 0_c_int, int(nrec, c_int), rec))
            do k = 1, nrec
               it = nint(rec(1, k))
               ist = nint(rec(12, k))
               test3 = huge(one)
               if (rec(7, k) /= 0.0_c_double) test3 = real(1.0_c_double/rec(7, k), wp)
               show = (n <= 40) .or. (it <= 10) .or. (it >= itnlim - 10) .or. (mod(it, 10) == 0) .or. &
                      (test3 <= 2.0_wp*ctol) .or. (rec(5, k) <= 10.0_wp*atol) .or. &
                      (rec(4, k) <= 10.0_wp*rec(13, k)) .or. (ist /= 0)
               if (show) write (nout, iter_fmt) it, rec(2:11, k)
            end do
         end if
      end if
      call log_exit(nout, istop, itn, anorm, acond, real(ex(1), wp), xnorm, rnorm, arnorm, real(ex(2), wp), &
                    nint(ex(3)))
   end subroutine lsqr_print_device_log

   ! =========================================================================================
   !  log text shared by the GPU path and the host path
   ! =========================================================================================

   subroutine log_header(nout, m, n, damp, wantse, atol, btol, conlim, itnlim)
      integer, intent(in) :: nout, m, n, itnlim
      real(wp), intent(in) :: damp, atol, btol, conlim
      logical, intent(in) :: wantse
      write (nout, '(//A)') enter_tag//'     Least-squares solution of  Ax = b'
      write (nout, '(A,I7,A,I7,A)') ' The matrix  A  has', m, ' rows   and', n, ' columns'
      write (nout, '(1P,A,E22.14,3X,A,L10)') ' damp   =', damp, 'wantse =', wantse
      write (nout, '(1P,A,E10.2,15x,A,E10.2)') ' atol   =', atol, 'conlim =', conlim
      write (nout, '(1P,A,E10.2,15x,A,I10)') ' btol   =', btol, 'itnlim =', itnlim
   end subroutine log_header

   subroutine log_titles(nout, damped, beta, test2)
      integer, intent(in) :: nout
      logical, intent(in) :: damped
      real(wp), intent(in) :: beta, test2
      if (damped) then
         write (nout, '(//A)') '   Itn       x(1)           Function     Compatible   LS     Norm Abar Cond Abar'
      else
         write (nout, '(//A)') '   Itn       x(1)           Function     Compatible   LS        Norm A    Cond A'
      end if
      write (nout, '(80X,A)') '    phi    dknorm   dxk  alfa_opt'
      write (nout, iter_fmt) 0, zero, beta, one, test2
      write (nout, '(A)') ''
   end subroutine log_titles

   subroutine log_exit(nout, istop, itn, anorm, acond, bnorm, xnorm, rnorm, arnorm, dxmax, maxdx)
      integer, intent(in) :: nout, istop, itn, maxdx
      real(wp), intent(in) :: anorm, acond, bnorm, xnorm, rnorm, arnorm, dxmax
      write (nout, '(//A,5X,A,I2,15X,A,I8)') exit_tag, 'istop  =', istop, 'itn    =', itn
      write (nout, '(1P,A,5X,A,E12.5,5X,A,E12.5)') exit_tag, 'anorm  =', anorm, 'acond  =', acond
      write (nout, '(1P,A,5X,A,E12.5,5X,A,E12.5)') exit_tag, 'bnorm  =', bnorm, 'xnorm  =', xnorm
      write (nout, '(1P,A,5X,A,E12.5,5X,A,E12.5)') exit_tag, 'rnorm  =', rnorm, 'arnorm =', arnorm
      write (nout, '(1P,A,5X,A,E8.1,A,I8)') exit_tag, 'max dx =', dxmax, ' occurred at itn ', maxdx
      write (nout, '(1P,A,5X,A,E8.1,A)') exit_tag, '       =', dxmax/(xnorm + 1.0e-20_wp), '*xnorm'
      write (nout, '(A,5X,A)') exit_tag, stop_msg(istop)
   end subroutine log_exit

   ! =========================================================================================
   !  lsqr_solver : host path around a user-written aprod
   ! =========================================================================================

   !> sqrt(a^2 + b^2) guarded against overflow (replaces src/lsqr.f90:1164-1179).
   pure function d2norm(a, b) result(r)
      real(wp), intent(in) :: a, b
      real(wp) :: r, s
      s = abs(a) + abs(b)
      r = zero
      if (s /= zero) r = s*sqrt((a/s)**2 + (b/s)**2)
   end function d2norm

   !> Paige & Saunders' LSQR around the user's operator (replaces src/lsqr.f90:432-882).
   !! Same dummy arguments and meaning as the reference.  The work is organised as
   !! (i) one Golub-Kahan step, (ii) the two plane rotations, (iii) the vector update,
   !! (iv) the norm estimates and the stopping decision.
   subroutine lsqr(me, m, n, damp, wantse, u, v, w, x, se, atol, btol, conlim, itnlim, nout, &
                   istop, itn, anorm, acond, rnorm, arnorm, xnorm)
      class(lsqr_solver), intent(inout) :: me
      integer, intent(in) :: m, n
      real(wp), intent(in) :: damp
      logical, intent(in) :: wantse
      real(wp), intent(inout) :: u(m), v(n), w(n)
      real(wp), intent(out) :: x(n)
      real(wp), dimension(*), intent(out) :: se
      real(wp), intent(in) :: atol, btol, conlim
      integer, intent(in) :: itnlim, nout
      integer, intent(out) :: istop, itn
      real(wp), intent(out) :: anorm, acond, rnorm, arnorm, xnorm

      logical :: damped, show
      integer :: nstop, maxdx
      real(wp) :: alpha, beta, bnorm, ctol, rhobar, phibar, psi, res2, dnorm, dxmax
      real(wp) :: xnorm1, cs2, sn2, z
      real(wp) :: rhbar1, cs1, sn1, rho, cs, sn, theta, phi, tau, t1, t2, t3
      real(wp) :: dknorm, dxk, delta, gambar, rhs, zbar, gamma
      real(wp) :: alfopt, test1, test2, test3, rtol, tmp
      integer, parameter :: nconv = 1

      if (nout /= 0) call log_header(nout, m, n, damp, wantse, atol, btol, conlim, itnlim)

      damped = damp > zero
      ctol = zero
      if (conlim > zero) ctol = one/conlim
      itn = 0; istop = 0; nstop = 0; maxdx = 0
      anorm = zero; acond = zero; dnorm = zero; dxmax = zero; res2 = zero; psi = zero
      xnorm = zero; xnorm1 = zero; cs2 = -one; sn2 = zero; z = zero
      bnorm = zero; rnorm = zero

      ! first Golub-Kahan vectors:  beta u = b,  alpha v = A'u
      v = zero
      x = zero
      if (wantse) se(1:n) = zero
      alpha = zero
      beta = dnrm2(m, u, 1)
      if (beta > zero) then
         call dscal(m, one/beta, u, 1)
         call me%aprod(2, m, n, v, u)
         alpha = dnrm2(n, v, 1)
      end if
      if (alpha > zero) then
         call dscal(n, one/alpha, v, 1)
         call dcopy(n, v, 1, w, 1)
      end if
      arnorm = alpha*beta
      bnorm = beta       ! defined even if no iteration runs (the reference leaves them unset)
      rnorm = beta

      if (arnorm /= zero) then
         rhobar = alpha
         phibar = beta
         if (nout /= 0) call log_titles(nout, damped, beta, alpha/beta)

         iterate: do
            itn = itn + 1

            ! (i)  beta u = A v - alpha u ;  alpha v = A'u - beta v
            call dscal(m, -alpha, u, 1)
            call me%aprod(1, m, n, v, u)
            beta = dnrm2(m, u, 1)
            anorm = d2norm(anorm, d2norm(d2norm(alpha, beta), damp))
            if (beta > zero) then
               call dscal(m, one/beta, u, 1)
               call dscal(n, -beta, v, 1)
               call me%aprod(2, m, n, v, u)
               alpha = dnrm2(n, v, 1)
               if (alpha > zero) call dscal(n, one/alpha, v, 1)
            end if

            ! (ii) rotation 1 removes damp, rotation 2 removes beta
            rhbar1 = rhobar
            if (damped) then
               rhbar1 = d2norm(rhobar, damp)
               cs1 = rhobar/rhbar1
               sn1 = damp/rhbar1
               psi = sn1*phibar
               phibar = cs1*phibar
            end if
            rho = d2norm(rhbar1, beta)
            cs = rhbar1/rho
            sn = beta/rho
            theta = sn*alpha
            rhobar = -cs*alpha
            phi = cs*phibar
            phibar = sn*phibar
            tau = sn*phi

            ! (iii) x and w (and the standard-error accumulators)
            t1 = phi/rho
            t2 = -theta/rho
            t3 = one/rho
            dknorm = sum((t3*w)**2)
            if (wantse) se(1:n) = se(1:n) + (t3*w)**2
            x = x + t1*w
            w = v + t2*w

            ! (iv) estimates of norm(d_k), norm(x), cond(Abar), norm(rbar), norm(Abar'rbar)
            dknorm = sqrt(dknorm)
            dnorm = d2norm(dnorm, dknorm)
            dxk = abs(phi*dknorm)
            if (dxmax < dxk) then
               dxmax = dxk
               maxdx = itn
            end if
            delta = sn2*rho
            gambar = -cs2*rho
            rhs = phi - delta*z
            zbar = rhs/gambar
            xnorm = d2norm(xnorm1, zbar)
            gamma = d2norm(gambar, theta)
            cs2 = gambar/gamma
            sn2 = theta/gamma
            z = rhs/gamma
            xnorm1 = d2norm(xnorm1, z)

            acond = anorm*dnorm
            res2 = d2norm(res2, psi)
            rnorm = d2norm(res2, phibar)
            arnorm = alpha*abs(tau)
            alfopt = sqrt(rnorm/(dnorm*xnorm))
            test1 = rnorm/bnorm
            test2 = zero
            if (rnorm > zero) test2 = arnorm/(anorm*rnorm)
            test3 = one/acond
            tmp = test1/(one + anorm*xnorm/bnorm)
            rtol = btol + atol*anorm*xnorm/bnorm

            ! stopping rules: machine-precision guards first, then the user's tolerances;
            ! later assignments win, so the precedence is 1 > 2 > 4 > 5
            if (itn >= itnlim) istop = 5
            if (one + test3 <= one) istop = 4
            if (one + test2 <= one) istop = 2
            if (one + tmp <= one) istop = 1
            if (test3 <= ctol) istop = 4
            if (test2 <= atol) istop = 2
            if (test1 <= rtol) istop = 1

            if (nout /= 0) then
               show = (n <= 40) .or. (itn <= 10) .or. (itn >= itnlim - 10) .or. (mod(itn, 10) == 0) .or. &
                      (test3 <= 2.0_wp*ctol) .or. (test2 <= 10.0_wp*atol) .or. (test1 <= 10.0_wp*rtol) .or. &
                      (istop /= 0)
               if (show) write (nout, iter_fmt) itn, x(1), rnorm, test1, test2, anorm, acond, phi, dknorm, dxk, alfopt
            end if

            ! the criteria must hold on nconv consecutive iterations
            if (istop == 0) then
               nstop = 0
            else
               nstop = nstop + 1
               if (nstop < nconv .and. itn < itnlim) istop = 0
            end if
            if (istop /= 0) exit iterate
         end do iterate

         if (wantse) then
            tmp = one
            if (m > n) tmp = real(m - n, wp)
            if (damped) tmp = real(m, wp)
            tmp = rnorm/sqrt(tmp)
            se(1:n) = tmp*sqrt(se(1:n))
         end if
      end if

      if (damped .and. istop == 2) istop = 3
      if (nout /= 0) call log_exit(nout, istop, itn, anorm, acond, bnorm, xnorm, rnorm, arnorm, dxmax, maxdx)
   end subroutine lsqr

   !> Do mode 1 and mode 2 of `aprod` describe the same matrix?  (replaces src/lsqr.f90:908-994)
   !! Tests y'(y + A x) = x'(x + A'y) on two fixed unit vectors.
   subroutine acheck(me, m, n, nout, eps, v, w, x, y, inform)
      class(lsqr_solver), intent(inout) :: me
      integer, intent(in) :: m, n, nout
      integer, intent(out) :: inform
      real(wp), intent(in) :: eps
      real(wp) :: v(n), w(m), x(n), y(m)
      real(wp) :: ywdot, xvdot, gap
      integer :: i

      if (nout /= 0) write (nout, '(//A)') 'Enter acheck. Test of aprod for LSQR and CRAIG'
      x = [(sqrt(real(i + 1, wp)), i=1, n)]
      y = [(one/sqrt(real(i + 1, wp)), i=1, m)]
      call dscal(n, one/dnrm2(n, x, 1), x, 1)
      call dscal(m, one/dnrm2(m, y, 1), y, 1)
      call dcopy(m, y, 1, w, 1)
      call dcopy(n, x, 1, v, 1)
      call me%aprod(1, m, n, x, w)          ! w = y + A x
      call me%aprod(2, m, n, v, y)          ! v = x + A'y
      ywdot = ddot(m, y, 1, w, 1)
      xvdot = ddot(n, x, 1, v, 1)
      gap = abs(ywdot - xvdot)/(one + abs(ywdot) + abs(xvdot))
      call acheck_verdict(nout, eps, gap, inform)
   end subroutine acheck

   !> the decision and the two report lines of `acheck` (src/lsqr.f90:984-992), shared by the host and the device form
   subroutine acheck_verdict(nout, eps, gap, inform)
      integer, intent(in) :: nout
      real(wp), intent(in) :: eps, gap
      integer, intent(out) :: inform
      real(wp), parameter :: power = 0.5_wp
      if (gap <= eps**power) then
         inform = 0
         if (nout /= 0) write (nout, '(1P,A,1X,E10.1)') 'aprod seems OK. Relative error =', gap
      else
         inform = 1
         if (nout /= 0) write (nout, '(1P,A,1X,E10.1)') 'aprod seems incorrect. Relative error =', gap
      end if
   end subroutine acheck_verdict

   !> `acheck` of an `lsqr_solver_ez`: the whole test -- the two vectors, both products, both dot products -- on the
   !! device operator (lsqrhip_acheck, include/lsqrhip.h), one call, no vector crosses PCIe.  v, w, x, y are the
   !! reference's work arrays: not touched.  A handle sharded over several GPUs has no device form of the test and takes
   !! the inherited host path over `aprod_ez`.
   subroutine acheck_ez(me, m, n, nout, eps, v, w, x, y, inform)
      class(lsqr_solver_ez), intent(inout) :: me
      integer, intent(in) :: m, n, nout
      integer, intent(out) :: inform
      real(wp), intent(in) :: eps
      real(wp) :: v(n), w(m), x(n), y(m)
      integer(c_int) :: inf
      real(c_double) :: err
      if (m /= me%m .or. n /= me%n .or. .not. c_associated(me%handle)) call check(4_c_int)
      if (me%sharded) then
         call acheck(me, m, n, nout, eps, v, w, x, y, inform)
         return
      end if
      if (nout /= 0) write (nout, '(//A)') 'Enter acheck. Test of aprod for LSQR and CRAIG'
      if (me%io32) then
         call check(lsqrhip_acheck_f32(me%handle, real(eps, c_double), inf, err))
      else
         call check(lsqrhip_acheck(me%handle, real(eps, c_double), inf, err))
      end if
      call acheck_verdict(nout, eps, real(err, wp), inform)
   end subroutine acheck_ez

   !> Which of Ax=b, min|Ax-b|, damped least squares does x solve?  (replaces src/lsqr.f90:1015-1154)
   subroutine xcheck(me, m, n, nout, anorm, damp, eps, b, u, v, w, x, inform, test1, test2, test3)
      class(lsqr_solver), intent(inout) :: me
      integer, intent(in) :: m, n, nout
      integer, intent(out) :: inform
      real(wp), intent(in) :: anorm, damp, eps
      real(wp), intent(out) :: test1, test2, test3
      real(wp), intent(in) :: b(m)
      real(wp), intent(out) :: u(m), v(n), w(n)
      real(wp), intent(in) :: x(n)
      real(wp), dimension(n) :: xwork

      xwork = x
      u = -b                                  ! r = b - A x, formed as -(-b + A x)
      call me%aprod(1, m, n, xwork, u)
      u = -u
      v = zero
      call me%aprod(2, m, n, v, u)            ! v = A'r
      w = v
      if (damp /= zero) w = w - damp**2*x     ! w = A'r - damp^2 x
      call xcheck_report(m, n, nout, anorm, damp, eps, b, u, v, w, x, inform, test1, test2, test3)
   end subroutine xcheck

   !> the norms, the three tests and the report of `xcheck` (src/lsqr.f90:1098-1154) from r = b - A x, A'r and
   !! A'r - damp^2 x -- shared by the host form (which gets them through `aprod`) and the device form
   subroutine xcheck_report(m, n, nout, anorm, damp, eps, b, u, v, w, x, inform, test1, test2, test3)
      integer, intent(in) :: m, n, nout
      integer, intent(out) :: inform
      real(wp), intent(in) :: anorm, damp, eps
      real(wp), intent(out) :: test1, test2, test3
      real(wp), intent(in) :: b(m), u(m), v(n), w(n), x(n)
      real(wp), parameter :: power = 0.5_wp
      real(wp) :: bnorm, xnorm, rho1, rho2, sigma1, sigma2, tol, dampsq

      dampsq = damp**2
      tol = eps**power
      bnorm = dnrm2(m, b, 1)
      xnorm = dnrm2(n, x, 1)
      rho1 = dnrm2(m, u, 1)
      sigma1 = dnrm2(n, v, 1)
      if (nout /= 0) then
         write (nout, '(//A)') 'Enter xcheck. Does x solve Ax = b, etc?'
         write (nout, '(1P,A,E10.3)') ' damp            =', damp
         write (nout, '(1P,A,E10.3)') ' norm(x)         =', xnorm
         write (nout, '(1P,A,E15.8,A)') ' norm(r)         =', rho1, ' = rho1'
         write (nout, '(1P,A,E10.3,5X,A)') ' norm(A''r)       =', sigma1, ' = sigma1'
      end if
      if (damp == zero) then
         rho2 = rho1
         sigma2 = sigma1
      else
         rho2 = sqrt(rho1**2 + dampsq*xnorm**2)
         sigma2 = dnrm2(n, w, 1)
         if (nout /= 0) then
            write (nout, '(1P/A,E10.3)') ' norm(s)         =', rho1/damp
            write (nout, '(1P,A,E10.3)') ' norm(x,s)       =', rho2/damp
            write (nout, '(1P,A,E15.8,A)') ' norm(rbar)      =', rho2, ' = rho2'
            write (nout, '(1P,A,E10.3,5X,A)') ' norm(Abar''rbar) =', sigma2, ' = sigma2'
         end if
      end if

      if (bnorm == zero .and. xnorm == zero) then
         inform = 0
         test1 = zero; test2 = zero; test3 = zero
      else
         inform = 4
         test1 = rho1/(bnorm + anorm*xnorm)
         test2 = zero
         if (rho1 > zero) test2 = sigma1/(anorm*rho1)
         test3 = test2
         if (rho2 > zero) test3 = sigma2/(anorm*rho2)
         if (test3 <= tol) inform = 3
         if (test2 <= tol) inform = 2
         if (test1 <= tol) inform = 1
      end if
      if (nout /= 0) then
         write (nout, '(/A,I2)') ' inform          =', inform
         write (nout, '(1P,A,E10.3)') ' tol             =', tol
         write (nout, '(1P,A,E10.3,A)') ' test1           =', test1, ' (Ax = b)'
         write (nout, '(1P,A,E10.3,A)') ' test2           =', test2, ' (least-squares)'
         write (nout, '(1P,A,E10.3,A)') ' test3           =', test3, ' (damped least-squares)'
      end if
   end subroutine xcheck_report

   !> `xcheck` of an `lsqr_solver_ez`: r = b - A x, A'r and A'r - damp^2 x by the device operator in ONE call
   !! (lsqrhip_xcheck: b and x go up once, u, v, w come back once -- the inherited form ships both vectors up and down
   !! for each of its two products); the norms and the report are the host form's, from the returned vectors.
   subroutine xcheck_ez(me, m, n, nout, anorm, damp, eps, b, u, v, w, x, inform, test1, test2, test3)
      class(lsqr_solver_ez), intent(inout) :: me
      integer, intent(in) :: m, n, nout
      integer, intent(out) :: inform
      real(wp), intent(in) :: anorm, damp, eps
      real(wp), intent(out) :: test1, test2, test3
      real(wp), intent(in) :: b(m)
      real(wp), intent(out) :: u(m), v(n), w(n)
      real(wp), intent(in) :: x(n)
      integer(c_int) :: inf
      real(c_double) :: tests(3)
      real(c_double), allocatable :: bl(:), xl(:), ul(:), vl(:), wl(:)
      real(c_float), allocatable :: bf(:), xf(:), uf(:), vf(:), wf(:)
      if (m /= me%m .or. n /= me%n .or. .not. c_associated(me%handle)) call check(4_c_int)
      if (me%sharded) then
         call xcheck(me, m, n, nout, anorm, damp, eps, b, u, v, w, x, inform, test1, test2, test3)
         return
      end if
      if (me%io32) then
         allocate (bf(max(m, 1)), xf(max(n, 1)), uf(max(m, 1)), vf(max(n, 1)), wf(max(n, 1)))
         bf(1:m) = real(b, c_float)
         xf(1:n) = real(x, c_float)
         call check(lsqrhip_xcheck_f32(me%handle, real(anorm, c_double), real(damp, c_double), real(eps, c_double), &
                                       bf, xf, uf, vf, wf, inf, tests))
         u = real(uf(1:m), wp)
         v = real(vf(1:n), wp)
         w = real(wf(1:n), wp)
      else
         allocate (bl(max(m, 1)), xl(max(n, 1)), ul(max(m, 1)), vl(max(n, 1)), wl(max(n, 1)))
         bl(1:m) = b
         xl(1:n) = x
         call check(lsqrhip_xcheck(me%handle, real(anorm, c_double), real(damp, c_double), real(eps, c_double), &
                                   bl, xl, ul, vl, wl, inf, tests))
         u = real(ul(1:m), wp)
         v = real(vl(1:n), wp)
         w = real(wl(1:n), wp)
      end if
      call xcheck_report(m, n, nout, anorm, damp, eps, b, u, v, w, x, inform, test1, test2, test3)
   end subroutine xcheck_ez

   !> `lsqr` of an `lsqr_solver_ez` (src/lsqr.f90:432-882 called on the EZ type): the device solve with the CALL's
   !! tolerances, limits and log unit (not the object's), b = u on entry.  The inherited host loop would ship both
   !! vectors over PCIe twice per iteration through `aprod_ez`.  u, v, w are the reference's work arrays: left as they
   !! are (the reference leaves its last bidiagonalisation vectors there, which nothing can use).
   subroutine lsqr_ez(me, m, n, damp, wantse, u, v, w, x, se, atol, btol, conlim, itnlim, nout, &
                      istop, itn, anorm, acond, rnorm, arnorm, xnorm)
      class(lsqr_solver_ez), intent(inout) :: me
      integer, intent(in) :: m, n
      real(wp), intent(in) :: damp
      logical, intent(in) :: wantse
      real(wp), intent(inout) :: u(m), v(n), w(n)
      real(wp), intent(out) :: x(n)
      real(wp), dimension(*), intent(out) :: se
      real(wp), intent(in) :: atol, btol, conlim
      integer, intent(in) :: itnlim, nout
      integer, intent(out) :: istop, itn
      real(wp), intent(out) :: anorm, acond, rnorm, arnorm, xnorm
      real(c_double), allocatable :: xl(:), sel(:), bl(:)
      real(c_float), allocatable :: xf(:), sef(:), bf(:)
      integer(c_int) :: istop_, itn_, ws, wl
      real(c_double) :: sc(5)
      if (m /= me%m .or. n /= me%n .or. .not. c_associated(me%handle)) call check(4_c_int)
      ws = merge(1_c_int, 0_c_int, wantse)
      wl = merge(1_c_int, 0_c_int, nout /= 0)
      if (me%io32) then
         allocate (xf(max(n, 1)), sef(max(n, 1)), bf(max(m, 1)))
         bf(1:m) = real(u, c_float)
         call check(lsqrhip_solve_f32(me%handle, bf, real(damp, c_double), real(atol, c_double), real(btol, c_double), &
                                      real(conlim, c_double), int(itnlim, c_int), ws, wl, xf, sef, istop_, itn_, &
                                      sc(1), sc(2), sc(3), sc(4), sc(5)))
         x = real(xf(1:n), wp)
         if (wantse) se(1:n) = real(sef(1:n), wp)
      else
         allocate (xl(max(n, 1)), sel(max(n, 1)), bl(max(m, 1)))
         bl(1:m) = u
         call check(lsqrhip_solve(me%handle, bl, real(damp, c_double), real(atol, c_double), real(btol, c_double), &
                                  real(conlim, c_double), int(itnlim, c_int), ws, wl, xl, sel, istop_, itn_, &
                                  sc(1), sc(2), sc(3), sc(4), sc(5)))
         x = real(xl(1:n), wp)
         if (wantse) se(1:n) = real(sel(1:n), wp)
      end if
      istop = istop_
      itn = itn_
      anorm = real(sc(1), wp); acond = real(sc(2), wp); rnorm = real(sc(3), wp)
      arnorm = real(sc(4), wp); xnorm = real(sc(5), wp)
      if (nout /= 0) call lsqr_print_device_log(me%handle, nout, m, n, damp, wantse, atol, btol, conlim, itnlim, &
                                                istop, itn, anorm, acond, rnorm, arnorm, xnorm)
      if (.false.) v(1) = w(1)
   end subroutine lsqr_ez

end module lsqr_module
