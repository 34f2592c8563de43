﻿!mod$ v1 sum:bfae6b404ba0ad73
!need$ 3138c98327cd2df8 n lsqr_kinds
module lsqpblas_module
use lsqr_kinds,only:wp
use lsqr_kinds,only:zero
use lsqr_kinds,only:one
private::wp
private::zero
private::one
private::first_index
contains
subroutine dcopy(n,dx,incx,dy,incy)
integer(4)::n
real(4)::dx(1_8:*)
integer(4)::incx
real(4)::dy(1_8:*)
integer(4)::incy
end
function ddot(n,dx,incx,dy,incy)
integer(4)::n
real(4)::dx(1_8:*)
integer(4)::incx
real(4)::dy(1_8:*)
integer(4)::incy
real(4)::ddot
end
function dnrm2(n,x,incx)
integer(4)::n
real(4)::x(1_8:*)
integer(4)::incx
real(4)::dnrm2
end
subroutine dscal(n,da,dx,incx)
integer(4)::n
real(4)::da
real(4)::dx(1_8:*)
integer(4)::incx
end
pure function first_index(n,inc)
integer(4),intent(in)::n
integer(4),intent(in)::inc
integer(4)::first_index
end
end
