!> Triggers ONE of the reference's `error stop` conditions (src/lsqr.f90:109-111, 152, 197),
!! selected by the first command-line argument, through the Fortran host layer.
!! tests/test_fortran.py checks the message text and the non-zero exit status.
program test_errors
   use lsqr_kinds
   use lsqr_module, only: lsqr_solver_ez
   implicit none
   character(len=32) :: which
   integer :: irow(9), icol(9), istop, i, j
   real(wp) :: a(9), b(3), x(3), y(3)
   type(lsqr_solver_ez) :: s
   a = real([1, 4, 7, 2, 5, 88, 3, 66, 9], wp)
   do j = 1, 3
      do i = 1, 3
         irow((j - 1)*3 + i) = i
         icol((j - 1)*3 + i) = j
      end do
   end do
   b = [1.0_wp, 2.0_wp, 3.0_wp]
   call get_command_argument(1, which)
   select case (trim(which))
   case ('sizes')        ! size(a) /= size(irow)
      call s%initialize(3, 3, a(1:8), irow, icol)
   case ('irow')         ! an index beyond m
      irow(5) = 4
      call s%initialize(3, 3, a, irow, icol)
   case ('icol')         ! an index beyond n
      icol(9) = 7
      call s%initialize(3, 3, a, irow, icol)
   case ('notinit')      ! aprod on an object that was never initialised
      x = 0; y = 0
      call s%aprod(1, 3, 3, x, y)
   case ('dims')         ! aprod with dimensions that are not the stored ones
      call s%initialize(3, 3, a, irow, icol)
      x = 0; y = 0
      call s%aprod(1, 4, 3, x, y)
   case ('mode')         ! aprod with mode 3
      call s%initialize(3, 3, a, irow, icol)
      x = 0; y = 0
      call s%aprod(3, 3, 3, x, y)
   case ('ok')
      call s%initialize(3, 3, a, irow, icol)
      call s%solve(b, zero, x, istop)
      write (*, '(A,I2)') 'OK istop=', istop
   case default
      error stop 'unknown selector'
   end select
   write (*, '(A)') 'NO ERROR RAISED'
end program test_errors
