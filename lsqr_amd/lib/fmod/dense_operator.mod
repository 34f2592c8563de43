﻿!mod$ v1 sum:01dc125599448983
!need$ e240aee1d6d8483a n lsqr_module
!need$ 56c4c5b6fa2ed0dc n lsqr_kinds
module dense_operator
use lsqr_module,only:lsqr_solver
use lsqr_kinds,only:wp
use lsqr_kinds,only:zero
use lsqr_kinds,only:one
type,extends(lsqr_solver)::dense_solver
real(8),allocatable::amat(:,:)
contains
procedure::aprod=>dense_aprod
end type
contains
subroutine dense_aprod(me,mode,m,n,x,y)
class(dense_solver),intent(inout)::me
integer(4),intent(in)::mode
integer(4),intent(in)::m
integer(4),intent(in)::n
real(8),intent(inout)::x(:)
real(8),intent(inout)::y(:)
end
end
