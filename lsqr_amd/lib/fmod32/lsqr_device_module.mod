﻿!mod$ v1 sum:88aefd23d3712bf0
!need$ 50ccce7e511721c5 n lsqr_module
!need$ 0bde2ac47243ead2 i iso_c_binding
!need$ bb381bf46e508468 i __fortran_builtins
!need$ 3138c98327cd2df8 n lsqr_kinds
module lsqr_device_module
use,intrinsic::__fortran_builtins,only:__builtin_c_ptr
use lsqr_module,only:lsqr_print_device_log
use lsqr_module,only:lsqr_check_status
use,intrinsic::iso_c_binding,only:c_associated
use,intrinsic::iso_c_binding,only:c_funloc
use,intrinsic::iso_c_binding,only:c_funptr
use,intrinsic::iso_c_binding,only:c_f_pointer
use,intrinsic::iso_c_binding,only:c_loc
use,intrinsic::iso_c_binding,only:c_null_funptr
use,intrinsic::iso_c_binding,only:c_null_ptr
use,intrinsic::iso_c_binding,only:c_ptr
use,intrinsic::iso_c_binding,only:c_sizeof
use,intrinsic::iso_c_binding,only:operator(==)
use,intrinsic::iso_c_binding,only:operator(/=)
use,intrinsic::iso_c_binding,only:c_int8_t
use,intrinsic::iso_c_binding,only:c_int16_t
use,intrinsic::iso_c_binding,only:c_int32_t
use,intrinsic::iso_c_binding,only:c_int64_t
use,intrinsic::iso_c_binding,only:c_int128_t
use,intrinsic::iso_c_binding,only:c_int
use,intrinsic::iso_c_binding,only:c_short
use,intrinsic::iso_c_binding,only:c_long
use,intrinsic::iso_c_binding,only:c_long_long
use,intrinsic::iso_c_binding,only:c_signed_char
use,intrinsic::iso_c_binding,only:c_size_t
use,intrinsic::iso_c_binding,only:c_intmax_t
use,intrinsic::iso_c_binding,only:c_intptr_t
use,intrinsic::iso_c_binding,only:c_ptrdiff_t
use,intrinsic::iso_c_binding,only:c_int_least8_t
use,intrinsic::iso_c_binding,only:c_int_fast8_t
use,intrinsic::iso_c_binding,only:c_int_least16_t
use,intrinsic::iso_c_binding,only:c_int_fast16_t
use,intrinsic::iso_c_binding,only:c_int_least32_t
use,intrinsic::iso_c_binding,only:c_int_fast32_t
use,intrinsic::iso_c_binding,only:c_int_least64_t
use,intrinsic::iso_c_binding,only:c_int_fast64_t
use,intrinsic::iso_c_binding,only:c_int_least128_t
use,intrinsic::iso_c_binding,only:c_int_fast128_t
use,intrinsic::iso_c_binding,only:c_float
use,intrinsic::iso_c_binding,only:c_double
use,intrinsic::iso_c_binding,only:c_long_double
use,intrinsic::iso_c_binding,only:c_float_complex
use,intrinsic::iso_c_binding,only:c_double_complex
use,intrinsic::iso_c_binding,only:c_long_double_complex
use,intrinsic::iso_c_binding,only:c_bool
use,intrinsic::iso_c_binding,only:c_char
use,intrinsic::iso_c_binding,only:c_null_char
use,intrinsic::iso_c_binding,only:c_alert
use,intrinsic::iso_c_binding,only:c_backspace
use,intrinsic::iso_c_binding,only:c_form_feed
use,intrinsic::iso_c_binding,only:c_new_line
use,intrinsic::iso_c_binding,only:c_carriage_return
use,intrinsic::iso_c_binding,only:c_horizontal_tab
use,intrinsic::iso_c_binding,only:c_vertical_tab
use,intrinsic::iso_c_binding,only:c_float128
use,intrinsic::iso_c_binding,only:c_float128_complex
use,intrinsic::iso_c_binding,only:c_uint8_t
use,intrinsic::iso_c_binding,only:c_uint16_t
use,intrinsic::iso_c_binding,only:c_uint32_t
use,intrinsic::iso_c_binding,only:c_uint64_t
use,intrinsic::iso_c_binding,only:c_uint128_t
use,intrinsic::iso_c_binding,only:c_unsigned_char
use,intrinsic::iso_c_binding,only:c_unsigned_short
use,intrinsic::iso_c_binding,only:c_unsigned
use,intrinsic::iso_c_binding,only:c_unsigned_long
use,intrinsic::iso_c_binding,only:c_unsigned_long_long
use,intrinsic::iso_c_binding,only:c_uintmax_t
use,intrinsic::iso_c_binding,only:c_uint_fast8_t
use,intrinsic::iso_c_binding,only:c_uint_fast16_t
use,intrinsic::iso_c_binding,only:c_uint_fast32_t
use,intrinsic::iso_c_binding,only:c_uint_fast64_t
use,intrinsic::iso_c_binding,only:c_uint_fast128_t
use,intrinsic::iso_c_binding,only:c_uint_least8_t
use,intrinsic::iso_c_binding,only:c_uint_least16_t
use,intrinsic::iso_c_binding,only:c_uint_least32_t
use,intrinsic::iso_c_binding,only:c_uint_least64_t
use,intrinsic::iso_c_binding,only:c_uint_least128_t
use,intrinsic::iso_c_binding,only:c_f_procpointer
use lsqr_kinds,only:wp
use lsqr_kinds,only:zero
use lsqr_kinds,only:one
use,intrinsic::__fortran_builtins,only:iso_c_binding$__fortran_builtins$c_associated_c_ptr=>c_associated_c_ptr
private::__builtin_c_ptr
private::lsqr_print_device_log
private::lsqr_check_status
private::c_associated
private::c_funloc
private::c_funptr
private::c_f_pointer
private::c_loc
private::c_null_funptr
private::c_null_ptr
private::c_ptr
private::c_sizeof
private::operator(==)
private::operator(/=)
private::c_int8_t
private::c_int16_t
private::c_int32_t
private::c_int64_t
private::c_int128_t
private::c_int
private::c_short
private::c_long
private::c_long_long
private::c_signed_char
private::c_size_t
private::c_intmax_t
private::c_intptr_t
private::c_ptrdiff_t
private::c_int_least8_t
private::c_int_fast8_t
private::c_int_least16_t
private::c_int_fast16_t
private::c_int_least32_t
private::c_int_fast32_t
private::c_int_least64_t
private::c_int_fast64_t
private::c_int_least128_t
private::c_int_fast128_t
private::c_float
private::c_double
private::c_long_double
private::c_float_complex
private::c_double_complex
private::c_long_double_complex
private::c_bool
private::c_char
private::c_null_char
private::c_alert
private::c_backspace
private::c_form_feed
private::c_new_line
private::c_carriage_return
private::c_horizontal_tab
private::c_vertical_tab
private::c_float128
private::c_float128_complex
private::c_uint8_t
private::c_uint16_t
private::c_uint32_t
private::c_uint64_t
private::c_uint128_t
private::c_unsigned_char
private::c_unsigned_short
private::c_unsigned
private::c_unsigned_long
private::c_unsigned_long_long
private::c_uintmax_t
private::c_uint_fast8_t
private::c_uint_fast16_t
private::c_uint_fast32_t
private::c_uint_fast64_t
private::c_uint_fast128_t
private::c_uint_least8_t
private::c_uint_least16_t
private::c_uint_least32_t
private::c_uint_least64_t
private::c_uint_least128_t
private::c_f_procpointer
private::wp
private::zero
private::one
private::iso_c_binding$__fortran_builtins$c_associated_c_ptr
type::lsqr_device_handle
type(c_ptr)::handle=__builtin_c_ptr(__address=0_8)
integer(4)::m=0_4
integer(4)::n=0_4
contains
procedure::lsqr=>lsqr_dev
procedure::acheck=>acheck_dev
procedure::xcheck=>xcheck_dev
procedure::destroy=>destroy_dev
end type
type,private::dev_box
class(lsqr_solver_device),pointer::p=>NULL()
end type
intrinsic::null
private::null
type,abstract,extends(lsqr_device_handle)::lsqr_solver_device
type(dev_box),pointer,private::box=>NULL()
contains
procedure(aprod_device_func),deferred::aprod_device
procedure::initialize_device
end type
type,extends(lsqr_device_handle)::lsqr_test_problem_device
real(4)::acond=0._4
real(4)::rnorm=0._4
real(4),allocatable::b(:)
real(4),allocatable::xtrue(:)
contains
procedure::create=>create_test_problem
end type
private::aprod_device_func
abstract interface
subroutine aprod_device_func(me,mode,m,n,x,y,stream)
import::c_ptr
import::lsqr_solver_device
class(lsqr_solver_device),intent(inout)::me
integer(4),intent(in)::mode
integer(4),intent(in)::m
integer(4),intent(in)::n
type(c_ptr),intent(in)::x
type(c_ptr),intent(in)::y
type(c_ptr),intent(in)::stream
end
end interface
private::lsqrhip_create_operator
interface
function lsqrhip_create_operator(m,n,aprod,user,h) bind(c,name="lsqrhip_create_operator") result(rc)
import::c_funptr
import::c_ptr
integer(4),value::m
integer(4),value::n
type(c_funptr),value::aprod
type(c_ptr),value::user
type(c_ptr),intent(out)::h
integer(4)::rc
end
end interface
private::lsqrhip_lstp_create
interface
function lsqrhip_lstp_create(m,n,nduplc,npower,damp,h,acond,rnorm) bind(c,name="lsqrhip_lstp_create") result(rc)
import::c_ptr
integer(4),value::m
integer(4),value::n
integer(4),value::nduplc
integer(4),value::npower
real(8),value::damp
type(c_ptr),intent(out)::h
real(8),intent(out)::acond
real(8),intent(out)::rnorm
integer(4)::rc
end
end interface
private::lsqrhip_lstp_vectors
interface
function lsqrhip_lstp_vectors(h,xtrue,b,d,hy,hz,d_b) bind(c,name="lsqrhip_lstp_vectors") result(rc)
import::c_ptr
type(c_ptr),value::h
real(8),intent(out)::xtrue(1_8:*)
real(8),intent(out)::b(1_8:*)
type(c_ptr),value::d
type(c_ptr),value::hy
type(c_ptr),value::hz
type(c_ptr),value::d_b
integer(4)::rc
end
end interface
private::lsqrhip_destroy
interface
function lsqrhip_destroy(h) bind(c,name="lsqrhip_destroy") result(rc)
import::c_ptr
type(c_ptr),value::h
integer(4)::rc
end
end interface
private::lsqrhip_solve
interface
function lsqrhip_solve(h,b,damp,atol,btol,conlim,itnlim,wantse,want_log,x,se,istop,itn,anorm,acond,rnorm,arnorm,xnorm) bind(c,name="lsqrhip_solve") result(rc)
import::c_ptr
type(c_ptr),value::h
real(8),intent(in)::b(1_8:*)
real(8),value::damp
real(8),value::atol
real(8),value::btol
real(8),value::conlim
integer(4),value::itnlim
integer(4),value::wantse
integer(4),value::want_log
real(8),intent(out)::x(1_8:*)
real(8),intent(inout)::se(1_8:*)
integer(4),intent(out)::istop
integer(4),intent(out)::itn
real(8),intent(out)::anorm
real(8),intent(out)::acond
real(8),intent(out)::rnorm
real(8),intent(out)::arnorm
real(8),intent(out)::xnorm
integer(4)::rc
end
end interface
private::lsqrhip_acheck
interface
function lsqrhip_acheck(h,eps,inform,relerr) bind(c,name="lsqrhip_acheck") result(rc)
import::c_ptr
type(c_ptr),value::h
real(8),value::eps
integer(4),intent(out)::inform
real(8),intent(out)::relerr
integer(4)::rc
end
end interface
private::lsqrhip_xcheck
interface
function lsqrhip_xcheck(h,anorm,damp,eps,b,x,u,v,w,inform,tests) bind(c,name="lsqrhip_xcheck") result(rc)
import::c_ptr
type(c_ptr),value::h
real(8),value::anorm
real(8),value::damp
real(8),value::eps
real(8),intent(in)::b(1_8:*)
real(8),intent(in)::x(1_8:*)
real(8),intent(out)::u(1_8:*)
real(8),intent(out)::v(1_8:*)
real(8),intent(out)::w(1_8:*)
integer(4),intent(out)::inform
real(8),intent(out)::tests(1_8:3_8)
integer(4)::rc
end
end interface
interface
function lsqrhip_aprod_device(h,mode,d_x,d_y) bind(c,name="lsqrhip_aprod_device") result(rc)
import::c_ptr
type(c_ptr),value::h
integer(4),value::mode
type(c_ptr),value::d_x
type(c_ptr),value::d_y
integer(4)::rc
end
end interface
interface
function lsqrhip_set_stream(h,stream) bind(c,name="lsqrhip_set_stream") result(rc)
import::c_ptr
type(c_ptr),value::h
type(c_ptr),value::stream
integer(4)::rc
end
end interface
private::trampoline
private::initialize_device
private::destroy_dev
private::create_test_problem
private::lsqr_dev
private::acheck_dev
private::xcheck_dev
contains
function trampoline(user,mode,m,n,d_x,d_y,stream) bind(c) result(rc)
type(c_ptr),value::user
integer(4),value::mode
integer(4),value::m
integer(4),value::n
type(c_ptr),value::d_x
type(c_ptr),value::d_y
type(c_ptr),value::stream
integer(4)::rc
end
subroutine initialize_device(me,m,n)
class(lsqr_solver_device),intent(inout),target::me
integer(4),intent(in)::m
integer(4),intent(in)::n
end
subroutine destroy_dev(me)
class(lsqr_device_handle),intent(inout)::me
end
subroutine create_test_problem(me,m,n,nduplc,npower,damp)
class(lsqr_test_problem_device),intent(inout)::me
integer(4),intent(in)::m
integer(4),intent(in)::n
integer(4),intent(in)::nduplc
integer(4),intent(in)::npower
real(4),intent(in)::damp
end
subroutine lsqr_dev(me,m,n,damp,wantse,u,v,w,x,se,atol,btol,conlim,itnlim,nout,istop,itn,anorm,acond,rnorm,arnorm,xnorm)
class(lsqr_device_handle),intent(inout)::me
integer(4),intent(in)::m
integer(4),intent(in)::n
real(4),intent(in)::damp
logical(4),intent(in)::wantse
real(4),intent(inout)::u(1_8:int(m,kind=8))
real(4),intent(inout)::v(1_8:int(n,kind=8))
real(4),intent(inout)::w(1_8:int(n,kind=8))
real(4),intent(out)::x(1_8:int(n,kind=8))
real(4),intent(inout)::se(1_8:*)
real(4),intent(in)::atol
real(4),intent(in)::btol
real(4),intent(in)::conlim
integer(4),intent(in)::itnlim
integer(4),intent(in)::nout
integer(4),intent(out)::istop
integer(4),intent(out)::itn
real(4),intent(out)::anorm
real(4),intent(out)::acond
real(4),intent(out)::rnorm
real(4),intent(out)::arnorm
real(4),intent(out)::xnorm
end
subroutine acheck_dev(me,m,n,nout,eps,v,w,x,y,inform)
class(lsqr_device_handle),intent(inout)::me
integer(4),intent(in)::m
integer(4),intent(in)::n
integer(4),intent(in)::nout
real(4),intent(in)::eps
real(4),intent(inout)::v(1_8:int(n,kind=8))
real(4),intent(inout)::w(1_8:int(m,kind=8))
real(4),intent(inout)::x(1_8:int(n,kind=8))
real(4),intent(inout)::y(1_8:int(m,kind=8))
integer(4),intent(out)::inform
end
subroutine xcheck_dev(me,m,n,nout,anorm,damp,eps,b,u,v,w,x,inform,test1,test2,test3)
class(lsqr_device_handle),intent(inout)::me
integer(4),intent(in)::m
integer(4),intent(in)::n
integer(4),intent(in)::nout
real(4),intent(in)::anorm
real(4),intent(in)::damp
real(4),intent(in)::eps
real(4),intent(in)::b(1_8:int(m,kind=8))
real(4),intent(out)::u(1_8:int(m,kind=8))
real(4),intent(out)::v(1_8:int(n,kind=8))
real(4),intent(out)::w(1_8:int(n,kind=8))
real(4),intent(in)::x(1_8:int(n,kind=8))
integer(4),intent(out)::inform
real(4),intent(out)::test1
real(4),intent(out)::test2
real(4),intent(out)::test3
end
end
