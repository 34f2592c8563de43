﻿!mod$ v1 sum:50ccce7e511721c5
!need$ 0bde2ac47243ead2 i iso_c_binding
!need$ bfae6b404ba0ad73 n lsqpblas_module
!need$ bb381bf46e508468 i __fortran_builtins
!need$ 3138c98327cd2df8 n lsqr_kinds
module lsqr_module
use,intrinsic::__fortran_builtins,only:__builtin_c_ptr
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
use lsqpblas_module,only:dcopy
use lsqpblas_module,only:ddot
use lsqpblas_module,only:dnrm2
use lsqpblas_module,only:dscal
use,intrinsic::__fortran_builtins,only:iso_c_binding$__fortran_builtins$c_associated_c_ptr=>c_associated_c_ptr
private::__builtin_c_ptr
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
private::dcopy
private::ddot
private::dnrm2
private::dscal
private::iso_c_binding$__fortran_builtins$c_associated_c_ptr
integer(4),parameter,private::log_stride=14_4
type,abstract::lsqr_solver
contains
procedure(aprod_func),deferred::aprod
procedure::lsqr
procedure::acheck
procedure::xcheck
end type
type,extends(lsqr_solver)::lsqr_solver_ez
integer(4),private::m=0_4
integer(4),private::n=0_4
integer(4),private::num_nonzero_elements=0_4
real(4),private::atol=0._4
real(4),private::btol=0._4
real(4),private::conlim=0._4
integer(4),private::itnlim=100_4
integer(4),private::nout=0_4
type(c_ptr),private::handle=__builtin_c_ptr(__address=0_8)
contains
procedure::initialize=>initialize_ez
procedure::solve=>solve_ez
procedure::aprod=>aprod_ez
procedure::destroy=>destroy_ez
procedure,private::copy_ez
generic::assignment(=)=>copy_ez
final::finalize_ez
end type
private::aprod_func
abstract interface
subroutine aprod_func(me,mode,m,n,x,y)
import::lsqr_solver
class(lsqr_solver),intent(inout)::me
integer(4),intent(in)::mode
integer(4),intent(in)::m
integer(4),intent(in)::n
real(4),intent(inout)::x(:)
real(4),intent(inout)::y(:)
end
end interface
private::lsqrhip_create
interface
function lsqrhip_create(m,n,nnz,irow,icol,a,h) bind(c,name="lsqrhip_create") result(rc)
import::c_ptr
integer(4),value::m
integer(4),value::n
integer(8),value::nnz
integer(4),intent(in)::irow(1_8:*)
integer(4),intent(in)::icol(1_8:*)
real(8),intent(in)::a(1_8:*)
type(c_ptr),intent(out)::h
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
private::lsqrhip_retain
interface
function lsqrhip_retain(h) bind(c,name="lsqrhip_retain") result(rc)
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
private::lsqrhip_aprod
interface
function lsqrhip_aprod(h,mode,x,y) bind(c,name="lsqrhip_aprod") result(rc)
import::c_ptr
type(c_ptr),value::h
integer(4),value::mode
real(8),intent(inout)::x(1_8:*)
real(8),intent(inout)::y(1_8:*)
integer(4)::rc
end
end interface
private::lsqrhip_log_count
interface
function lsqrhip_log_count(h) bind(c,name="lsqrhip_log_count") result(k)
import::c_ptr
type(c_ptr),value::h
integer(4)::k
end
end interface
private::lsqrhip_log_fetch
interface
function lsqrhip_log_fetch(h,first,count,records) bind(c,name="lsqrhip_log_fetch") result(rc)
import::c_ptr
type(c_ptr),value::h
integer(4),value::first
integer(4),value::count
real(8),intent(out)::records(1_8:*)
integer(4)::rc
end
end interface
private::lsqrhip_log_extras
interface
function lsqrhip_log_extras(h,ex) bind(c,name="lsqrhip_log_extras") result(rc)
import::c_ptr
type(c_ptr),value::h
real(8),intent(out)::ex(1_8:*)
integer(4)::rc
end
end interface
private::lsqrhip_last_error
interface
function lsqrhip_last_error() bind(c,name="lsqrhip_last_error") result(p)
import::c_ptr
type(c_ptr)::p
end
end interface
private::c_strlen
interface
function c_strlen(s) bind(c,name="strlen") result(k)
import::c_ptr
type(c_ptr),value::s
integer(8)::k
end
end interface
character(*,1),parameter,private::enter_tag=" Enter LSQR.  "
character(*,1),parameter,private::exit_tag=" Exit  LSQR.  "
character(*,1),parameter,private::iter_fmt="(1P, I6, 2E17.9, 4E10.2, E9.1, 3E8.1)"
character(53_4,1),parameter,private::stop_msg(0_8:5_8)=[CHARACTER(KIND=1,LEN=53)::"The exact solution is x = 0                          ","A solution to Ax = b was found, given atol, btol     ","A least-squares solution was found, given atol       ","A damped least-squares solution was found, given atol","Cond(Abar) seems to be too large, given conlim       ","The iteration limit was reached                      "]
private::check
private::initialize_ez
private::destroy_ez
private::finalize_ez
private::copy_ez
private::aprod_ez
private::solve_ez
private::print_device_log
private::log_header
private::log_titles
private::log_exit
private::d2norm
private::lsqr
private::acheck
private::xcheck
contains
subroutine check(rc)
integer(4),intent(in)::rc
end
subroutine lsqr_check_status(rc)
integer(4),intent(in)::rc
end
subroutine initialize_ez(me,m,n,a,irow,icol,atol,btol,conlim,itnlim,nout)
class(lsqr_solver_ez),intent(out)::me
integer(4),intent(in)::m
integer(4),intent(in)::n
real(4),intent(in)::a(:)
integer(4),intent(in)::irow(:)
integer(4),intent(in)::icol(:)
real(4),intent(in),optional::atol
real(4),intent(in),optional::btol
real(4),intent(in),optional::conlim
integer(4),intent(in),optional::itnlim
integer(4),intent(in),optional::nout
end
subroutine destroy_ez(me)
class(lsqr_solver_ez),intent(inout)::me
end
subroutine finalize_ez(me)
type(lsqr_solver_ez),intent(inout)::me
end
subroutine copy_ez(lhs,rhs)
class(lsqr_solver_ez),intent(inout)::lhs
class(lsqr_solver_ez),intent(in)::rhs
end
subroutine aprod_ez(me,mode,m,n,x,y)
class(lsqr_solver_ez),intent(inout)::me
integer(4),intent(in)::mode
integer(4),intent(in)::m
integer(4),intent(in)::n
real(4),intent(inout)::x(:)
real(4),intent(inout)::y(:)
end
subroutine solve_ez(me,b,damp,x,istop,se,itn,anorm,acond,rnorm,arnorm,xnorm)
class(lsqr_solver_ez),intent(inout)::me
real(4),intent(in)::b(1_8:int(me%m,kind=8))
real(4),intent(in)::damp
real(4),intent(out)::x(1_8:int(me%n,kind=8))
integer(4),intent(out)::istop
real(4),intent(out),optional::se(1_8:int(me%n,kind=8))
integer(4),intent(out),optional::itn
real(4),intent(out),optional::anorm
real(4),intent(out),optional::acond
real(4),intent(out),optional::rnorm
real(4),intent(out),optional::arnorm
real(4),intent(out),optional::xnorm
end
subroutine print_device_log(me,damp,wantse,istop,itn,anorm,acond,rnorm,arnorm,xnorm)
class(lsqr_solver_ez),intent(in)::me
real(4),intent(in)::damp
logical(4),intent(in)::wantse
integer(4),intent(in)::istop
integer(4),intent(in)::itn
real(4),intent(in)::anorm
real(4),intent(in)::acond
real(4),intent(in)::rnorm
real(4),intent(in)::arnorm
real(4),intent(in)::xnorm
end
subroutine lsqr_print_device_log(handle,nout,m,n,damp,wantse,atol,btol,conlim,itnlim,istop,itn,anorm,acond,rnorm,arnorm,xnorm)
type(c_ptr),intent(in)::handle
integer(4),intent(in)::nout
integer(4),intent(in)::m
integer(4),intent(in)::n
real(4),intent(in)::damp
logical(4),intent(in)::wantse
real(4),intent(in)::atol
real(4),intent(in)::btol
real(4),intent(in)::conlim
integer(4),intent(in)::itnlim
integer(4),intent(in)::istop
integer(4),intent(in)::itn
real(4),intent(in)::anorm
real(4),intent(in)::acond
real(4),intent(in)::rnorm
real(4),intent(in)::arnorm
real(4),intent(in)::xnorm
end
subroutine log_header(nout,m,n,damp,wantse,atol,btol,conlim,itnlim)
integer(4),intent(in)::nout
integer(4),intent(in)::m
integer(4),intent(in)::n
real(4),intent(in)::damp
logical(4),intent(in)::wantse
real(4),intent(in)::atol
real(4),intent(in)::btol
real(4),intent(in)::conlim
integer(4),intent(in)::itnlim
end
subroutine log_titles(nout,damped,beta,test2)
integer(4),intent(in)::nout
logical(4),intent(in)::damped
real(4),intent(in)::beta
real(4),intent(in)::test2
end
subroutine log_exit(nout,istop,itn,anorm,acond,bnorm,xnorm,rnorm,arnorm,dxmax,maxdx)
integer(4),intent(in)::nout
integer(4),intent(in)::istop
integer(4),intent(in)::itn
real(4),intent(in)::anorm
real(4),intent(in)::acond
real(4),intent(in)::bnorm
real(4),intent(in)::xnorm
real(4),intent(in)::rnorm
real(4),intent(in)::arnorm
real(4),intent(in)::dxmax
integer(4),intent(in)::maxdx
end
pure function d2norm(a,b) result(r)
real(4),intent(in)::a
real(4),intent(in)::b
real(4)::r
end
subroutine lsqr(me,m,n,damp,wantse,u,v,w,x,se,atol,btol,conlim,itnlim,nout,istop,itn,anorm,acond,rnorm,arnorm,xnorm)
class(lsqr_solver),intent(inout)::me
integer(4),intent(in)::m
integer(4),intent(in)::n
real(4),intent(in)::damp
logical(4),intent(in)::wantse
real(4),intent(inout)::u(1_8:int(m,kind=8))
real(4),intent(inout)::v(1_8:int(n,kind=8))
real(4),intent(inout)::w(1_8:int(n,kind=8))
real(4),intent(out)::x(1_8:int(n,kind=8))
real(4),intent(out)::se(1_8:*)
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
subroutine acheck(me,m,n,nout,eps,v,w,x,y,inform)
class(lsqr_solver),intent(inout)::me
integer(4),intent(in)::m
integer(4),intent(in)::n
integer(4),intent(in)::nout
real(4),intent(in)::eps
real(4)::v(1_8:int(n,kind=8))
real(4)::w(1_8:int(m,kind=8))
real(4)::x(1_8:int(n,kind=8))
real(4)::y(1_8:int(m,kind=8))
integer(4),intent(out)::inform
end
subroutine xcheck(me,m,n,nout,anorm,damp,eps,b,u,v,w,x,inform,test1,test2,test3)
class(lsqr_solver),intent(inout)::me
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
