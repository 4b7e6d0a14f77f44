! beom_cabi.f95 -- iso_c_binding view of include/beom_hip.h (libbeom_hip.so).
!
! This is the binding a maintainer of the reference adds to call the MI355X engine from
! Fortran: one bind(C) derived type mirroring `struct beom_params` and one interface per
! C entry point.  Arrays are handed over as C addresses (c_loc of the module arrays, or
! c_null_ptr where include/beom_hip.h allows NULL), strings as NUL-terminated
! character(kind=c_char) buffers of length lstr+1 (errm convention of shared_mod).
module beom_cabi
  use iso_c_binding
  implicit none
  public

  integer, parameter :: beom_max_layers = 16
  integer, parameter :: beom_abi_ver    = 2

  type, bind(C) :: beom_params
    integer(c_int32_t) :: abi_version, lm, mm, nlay, ndeg, nsal, variant, &
                          flag_nudging, dense_hint, slab_row0, slab_mm
    real(c_double)     :: dl, dt, grav, rho0, beta, epsi, gamm, del1, del2,  &
                          hmin, hsal, bvis, dvis, svis, bdrg, tdrg, qdrg,    &
                          hsbl, hbbl, g_fb, uadv, ocrp, rgld, mcbc, invf, w_ti
    real(c_double)     :: rhon(beom_max_layers)
  end type beom_params

  interface
    function beom_abi_version() bind(C, name = 'beom_abi_version') result(v)
      import :: c_int
      integer(c_int) :: v
    end function beom_abi_version

    function beom_create(prm, device, neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi, fcor,   &
                         h_th, h_to, nudg, fnud, hdot, tide, bodf, taus, handle,        &
                         errm, errm_len) bind(C, name = 'beom_create') result(rc)
      import :: c_int, c_ptr, c_char, beom_params
      type(beom_params), intent(in)  :: prm
      integer(c_int), value          :: device
      type(c_ptr), value             :: neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi, fcor, &
                                        h_th, h_to, nudg, fnud, hdot, tide, bodf, taus
      type(c_ptr), intent(out)       :: handle
      character(kind = c_char)       :: errm(*)
      integer(c_int), value          :: errm_len
      integer(c_int)                 :: rc
    end function beom_create

    function beom_destroy(handle) bind(C, name = 'beom_destroy') result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int)     :: rc
    end function beom_destroy

    function beom_upload_state(handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,    &
                               v_ll, tt3d, tb3d, tu3d, errm, errm_len)                   &
             bind(C, name = 'beom_upload_state') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,  &
                                  v_ll, tt3d, tb3d, tu3d
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_upload_state

    function beom_download_state(handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,  &
                                 v_ll, tt3d, tb3d, tu3d, errm, errm_len)                 &
             bind(C, name = 'beom_download_state') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,  &
                                  v_ll, tt3d, tb3d, tu3d
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_download_state

    function beom_step(handle, tstp_first, nsteps, tres, dtd8, dt_r, rsta, n_3d,        &
                       errm, errm_len) bind(C, name = 'beom_step') result(rc)
      import :: c_int, c_ptr, c_char, c_double
      type(c_ptr), value       :: handle
      integer(c_int), value    :: tstp_first, nsteps, n_3d
      real(c_double), value    :: tres, dtd8, dt_r, rsta
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_step

    function beom_sync(handle, errm, errm_len) bind(C, name = 'beom_sync') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_sync

    function beom_download_outputs(handle, h0r4, eta, u4, v4, minmax, thin_layer, errm, errm_len)      &
             bind(C, name = 'beom_download_outputs') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, h0r4, eta, u4, v4, minmax
      integer(c_int)           :: thin_layer
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_download_outputs

    function beom_multi_download_outputs(handle, h0r4, eta, u4, v4, minmax, thin_layer, errm, errm_len)  &
             bind(C, name = 'beom_multi_download_outputs') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, h0r4, eta, u4, v4, minmax
      integer(c_int)           :: thin_layer
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_multi_download_outputs

    function beom_multi_download_diag(handle, pvor, mont, v_cc, errm, errm_len)                          &
             bind(C, name = 'beom_multi_download_diag') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, pvor, mont, v_cc
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_multi_download_diag

    function beom_download_diag(handle, pvor, mont, v_cc, errm, errm_len)                                &
             bind(C, name = 'beom_download_diag') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, pvor, mont, v_cc
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_download_diag

    function beom_multi_set_open_boundaries(handle, nseg, segm, errm, errm_len)                          &
             bind(C, name = 'beom_multi_set_open_boundaries') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, segm
      integer(c_int), value    :: nseg
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_multi_set_open_boundaries

    function beom_set_rigid_lid(handle, Ow, Os, Osum_, pi_s, errm, errm_len)                              &
             bind(C, name = 'beom_set_rigid_lid') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, Ow, Os, Osum_, pi_s
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_set_rigid_lid

    function beom_set_open_boundaries(handle, nseg, segm, errm, errm_len)                                &
             bind(C, name = 'beom_set_open_boundaries') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, segm
      integer(c_int), value    :: nseg
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_set_open_boundaries

    ! ---- one process, several GPUs: row bands inside the library (include/beom_hip.h) ----
    function beom_multi_create(prm, ndev, devices, neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi, fcor,   &
                               h_th, h_to, nudg, fnud, hdot, tide, bodf, taus, handle,              &
                               errm, errm_len) bind(C, name = 'beom_multi_create') result(rc)
      import :: c_int, c_ptr, c_char, beom_params
      type(beom_params), intent(in)  :: prm
      integer(c_int), value          :: ndev
      integer(c_int), intent(in)     :: devices(*)
      type(c_ptr), value             :: neig, subc, mk_u, mk_v, mk_n, mkpe, mkpi, fcor, &
                                        h_th, h_to, nudg, fnud, hdot, tide, bodf, taus
      type(c_ptr), intent(out)       :: handle
      character(kind = c_char)       :: errm(*)
      integer(c_int), value          :: errm_len
      integer(c_int)                 :: rc
    end function beom_multi_create

    function beom_multi_destroy(handle) bind(C, name = 'beom_multi_destroy') result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int)     :: rc
    end function beom_multi_destroy

    function beom_multi_upload_state(handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,    &
                                     v_ll, tt3d, tb3d, tu3d, errm, errm_len)                   &
             bind(C, name = 'beom_multi_upload_state') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,  &
                                  v_ll, tt3d, tb3d, tu3d
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_multi_upload_state

    function beom_multi_download_state(handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,  &
                                       v_ll, tt3d, tb3d, tu3d, errm, errm_len)                 &
             bind(C, name = 'beom_multi_download_state') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle, hlay, u, v, h_u, h_v, rs_h, dmdx, dmdy, v_cc,  &
                                  v_ll, tt3d, tb3d, tu3d
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_multi_download_state

    function beom_multi_step(handle, tstp_first, nsteps, tres, dtd8, dt_r, rsta, n_3d,        &
                             errm, errm_len) bind(C, name = 'beom_multi_step') result(rc)
      import :: c_int, c_ptr, c_char, c_double
      type(c_ptr), value       :: handle
      integer(c_int), value    :: tstp_first, nsteps, n_3d
      real(c_double), value    :: tres, dtd8, dt_r, rsta
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_multi_step

    function beom_multi_sync(handle, errm, errm_len) bind(C, name = 'beom_multi_sync') result(rc)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value       :: handle
      character(kind = c_char) :: errm(*)
      integer(c_int), value    :: errm_len
      integer(c_int)           :: rc
    end function beom_multi_sync

    function beom_is_dense(handle) bind(C, name = 'beom_is_dense') result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int)     :: rc
    end function beom_is_dense
  end interface

contains

  ! C string (NUL-terminated) -> blank-padded Fortran string.
  function c_to_f(cbuf) result(s)
    character(kind = c_char), intent(in) :: cbuf(:)
    character(len = size(cbuf))          :: s
    integer :: k
    s = ' '
    do k = 1, size(cbuf)
      if ( cbuf(k) == c_null_char ) exit
      s(k:k) = cbuf(k)
    end do
  end function c_to_f

end module beom_cabi
