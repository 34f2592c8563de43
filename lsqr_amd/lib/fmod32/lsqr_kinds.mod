﻿!mod$ v1 sum:3138c98327cd2df8
!need$ f1de5abe9bfe2168 i iso_fortran_env
module lsqr_kinds
use,intrinsic::iso_fortran_env,only:real32
use,intrinsic::iso_fortran_env,only:real64
private::real32
private::real64
integer(4),parameter::wp=4_4
real(4),parameter::zero=0._4
real(4),parameter::one=1._4
end
