﻿!mod$ v1 sum:8587e60dcd189e35
!need$ f1de5abe9bfe2168 i iso_fortran_env
module lsqr_kinds
use,intrinsic::iso_fortran_env,only:real64
private::real64
integer(4),parameter::wp=8_4
real(8),parameter::zero=0._8
real(8),parameter::one=1._8
end
