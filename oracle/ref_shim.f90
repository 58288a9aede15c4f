! oracle/ref_shim.f90 -- TEST INFRASTRUCTURE, not product code.
!
! BIND(C) entry points over the *unmodified* reference leaf modules that
! compile on their own (no NetCDF, no PARAM_MOD):
!   gridcell_module.f90, interpolation_module.f90, point_in_polygon_module.f90,
!   random_module.f90, tension_module.f90      (under /root/reference/Model)
! The reference sources are compiled where they lie by oracle/Makefile into
! oracle/_ref/ (git-ignored).  This file is repo code: it only forwards
! arguments so that Python (ctypes) can pin oracle/ltrans_oracle.c against the
! real Fortran.
module ltrans_ref_shim
  use iso_c_binding
  implicit none
contains

  subroutine ref_init_genrand(seed) bind(C, name="ref_init_genrand")
    use random_mod, only: init_genrand
    integer(c_int), value :: seed
    call init_genrand(seed)
  end subroutine

  subroutine ref_genrand_int32(n, out) bind(C, name="ref_genrand_int32")
    use random_mod, only: genrand_int32
    integer(c_int), value :: n
    integer(c_int), intent(out) :: out(n)
    integer :: i
    do i = 1, n
      out(i) = genrand_int32()
    enddo
  end subroutine

  subroutine ref_genrand_real1(n, out) bind(C, name="ref_genrand_real1")
    use random_mod, only: genrand_real1
    integer(c_int), value :: n
    real(c_double), intent(out) :: out(n)
    integer :: i
    do i = 1, n
      out(i) = genrand_real1()
    enddo
  end subroutine

  subroutine ref_genrand_real3(n, out) bind(C, name="ref_genrand_real3")
    use random_mod, only: genrand_real3
    integer(c_int), value :: n
    real(c_double), intent(out) :: out(n)
    integer :: i
    do i = 1, n
      out(i) = genrand_real3()
    enddo
  end subroutine

  function ref_polintd(xa, ya, x) bind(C, name="ref_polintd") result(r)
    use int_mod, only: polintd
    real(c_double), intent(in) :: xa(3), ya(3)
    real(c_double), value :: x
    real(c_double) :: r
    r = polintd(xa, ya, 3, x)
  end function

  subroutine ref_linint(xa, ya, n, x, y, m) bind(C, name="ref_linint")
    use int_mod, only: linint
    integer(c_int), value :: n
    real(c_double), intent(in) :: xa(n), ya(n)
    real(c_double), value :: x
    real(c_double), intent(out) :: y, m
    call linint(xa, ya, n, x, y, m)
  end subroutine

  ! ele_x/ele_y are (4,elements) column-major, as in the reference.
  ! checkele <= 0 means "argument absent" (full scan).
  subroutine ref_gridcell(elements, ele_y, ele_x, xpos, ypos, p_ele, triangle, &
                          checkele) bind(C, name="ref_gridcell")
    use gridcell_mod, only: gridcell
    integer(c_int), value :: elements, checkele
    real(c_double), intent(in) :: ele_x(4, elements), ele_y(4, elements)
    real(c_double), value :: xpos, ypos
    integer(c_int), intent(inout) :: p_ele, triangle
    if (checkele > 0) then
      call gridcell(elements, ele_y, ele_x, xpos, ypos, p_ele, triangle, checkele)
    else
      call gridcell(elements, ele_y, ele_x, xpos, ypos, p_ele, triangle)
    endif
  end subroutine

  ! e is (n,2) column-major. onin: 1 = .TRUE., 0 = .FALSE., -1 = absent.
  function ref_inpoly(x, y, n, e, onin) bind(C, name="ref_inpoly") result(r)
    use pip_mod, only: inpoly
    real(c_double), value :: x, y
    integer(c_int), value :: n, onin
    real(c_double), intent(in) :: e(n, 2)
    integer(c_int) :: r
    logical :: l
    if (onin < 0) then
      l = inpoly(x, y, n, e)
    else
      l = inpoly(x, y, n, e, onin /= 0)
    endif
    r = 0
    if (l) r = 1
  end function

  subroutine ref_tspsi(n, x, y, yp, sigma, ier, sigerr) bind(C, name="ref_tspsi")
    use tension_mod, only: tspsi
    integer(c_int), value :: n
    real(c_double), intent(in) :: x(n), y(n)
    real(c_double), intent(inout) :: yp(n), sigma(n)
    integer(c_int), intent(out) :: ier
    integer(c_int), intent(inout) :: sigerr
    real(c_double) :: xx(n), yy(n)
    xx = x
    yy = y
    call tspsi(n, xx, yy, yp, sigma, ier, sigerr)
  end subroutine

  function ref_hval(t, n, x, y, yp, sigma, ier) bind(C, name="ref_hval") result(r)
    use tension_mod, only: hval
    real(c_double), value :: t
    integer(c_int), value :: n
    real(c_double), intent(in) :: x(n), y(n), yp(n), sigma(n)
    integer(c_int), intent(out) :: ier
    real(c_double) :: r
    real(c_double) :: tt, xx(n), yy(n), pp(n), ss(n)
    tt = t; xx = x; yy = y; pp = yp; ss = sigma
    r = hval(tt, n, xx, yy, pp, ss, ier)
  end function

  function ref_hpval(t, n, x, y, yp, sigma, ier) bind(C, name="ref_hpval") result(r)
    use tension_mod, only: hpval
    real(c_double), value :: t
    integer(c_int), value :: n
    real(c_double), intent(in) :: x(n), y(n), yp(n), sigma(n)
    integer(c_int), intent(out) :: ier
    real(c_double) :: r
    real(c_double) :: tt, xx(n), yy(n), pp(n), ss(n)
    tt = t; xx = x; yy = y; pp = yp; ss = sigma
    r = hpval(tt, n, xx, yy, pp, ss, ier)
  end function

  subroutine ref_snhcsh(x, sinhm, coshm, coshmm) bind(C, name="ref_snhcsh")
    use tension_mod, only: snhcsh
    real(c_double), value :: x
    real(c_double), intent(out) :: sinhm, coshm, coshmm
    real(c_double) :: xx
    xx = x
    call snhcsh(xx, sinhm, coshm, coshmm)
  end subroutine

  ! ---- timing loops (oracle/time_leaves.py): the reference's routines called nrep times from Fortran, so that what is
  ! timed is the routine and not the foreign-function call.  x, y, t hold nprof profiles / points that are cycled.
  ! acc (a sum of the results) keeps the work alive.
  subroutine ref_time_tspsi_hval(nrep, n, nprof, x, y, t, neval, acc) bind(C, name="ref_time_tspsi_hval")
    use tension_mod, only: tspsi, hval, hpval
    integer(c_int), value :: nrep, n, nprof, neval
    real(c_double), intent(in) :: x(n, nprof), y(n, nprof), t(neval, nprof)
    real(c_double), intent(out) :: acc
    real(c_double) :: xx(n), yy(n), yp(n), sigma(n), tt
    integer :: r, k, q, ier, sigerr
    acc = 0.0d0
    do r = 1, nrep
      q = mod(r - 1, nprof) + 1
      xx = x(:, q); yy = y(:, q)
      sigerr = 0
      call tspsi(n, xx, yy, yp, sigma, ier, sigerr)
      do k = 1, neval
        tt = t(k, q)
        if (neval > 1) acc = acc + hpval(tt, n, xx, yy, yp, sigma, ier)
        acc = acc + hval(tt, n, xx, yy, yp, sigma, ier)
      enddo
    enddo
  end subroutine

  subroutine ref_time_gridcell(nrep, elements, ele_y, ele_x, npts, px, py, acc) bind(C, name="ref_time_gridcell")
    use gridcell_mod, only: gridcell
    integer(c_int), value :: nrep, elements, npts
    real(c_double), intent(in) :: ele_x(4, elements), ele_y(4, elements), px(npts), py(npts)
    real(c_double), intent(out) :: acc
    integer :: r, q, p_ele, triangle, ce
    acc = 0.0d0
    do r = 1, nrep
      q = mod(r - 1, npts) + 1
      ce = mod(r - 1, elements) + 1
      p_ele = 0; triangle = 0
      call gridcell(elements, ele_y, ele_x, px(q), py(q), p_ele, triangle, ce)
      acc = acc + triangle
    enddo
  end subroutine

  subroutine ref_time_inpoly(nrep, n, e, npts, px, py, acc) bind(C, name="ref_time_inpoly")
    use pip_mod, only: inpoly
    integer(c_int), value :: nrep, n, npts
    real(c_double), intent(in) :: e(n, 2), px(npts), py(npts)
    real(c_double), intent(out) :: acc
    integer :: r, q
    acc = 0.0d0
    do r = 1, nrep
      q = mod(r - 1, npts) + 1
      if (inpoly(px(q), py(q), n, e)) acc = acc + 1.0d0
    enddo
  end subroutine

end module ltrans_ref_shim
