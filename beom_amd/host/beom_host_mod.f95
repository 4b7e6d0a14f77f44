! beom_host_mod.f95 -- Fortran-95 host for the MI355X engine: `module private_mod`.
!
! Drop-in for the reference's engine file under an UNCHANGED main.f95 and the user's
! shared_mod.f95: it exports the single symbol `run` (main.f95:28,34), takes every
! parameter from shared_mod (shared_mod.f95:41-111), reads the testcases/*.m input files
! and writes the output files of the reference's file contract, and hands the time loop
! to libbeom_hip.so through beom_cabi (iso_c_binding).  Written from scratch: allocatable
! state, own routine structure; citations `:NNNN` point at the reference routine
! (private_mod.f95) whose observable behaviour a block reproduces.
!
! rgld = 1 (the fork's rigid lid) runs on one device with ocrp = 1 (the only setting in which the reference
! initialises the lid's Poisson operators, :505-563); other combinations are refused with errc < 0 like any
! other bad option.  The fork's extra
! switches svis/tdrg/topt are not declared by the reference's own shared_mod.f95
! (SURVEY F2), so by default this host does not reference them (= 0).  With a shared_mod.f95
! that does declare them, compile with `-cpp -DBEOM_FORK_SWITCHES` (build_host.py does this
! by itself) and they are honoured: biharmonic viscosity, top drag, surface topography.
#ifdef BEOM_FORK_SWITCHES
#define FORK_SVIS svis
#define FORK_TDRG tdrg
#define FORK_TOPT topt
#else
#define FORK_SVIS 0._rw
#define FORK_TDRG 0._rw
#define FORK_TOPT 0._rw
#endif
module private_mod
  use shared_mod
  use iso_c_binding
  use beom_cabi
  implicit none
  private
  public run

  integer, parameter :: n1 = ndeg + 1             ! packed length incl. land sentinel 0

  integer(c_int32_t), allocatable, target :: neig(:,:), subc(:,:)
  real(c_double), allocatable, target ::                                              &
    mk_u(:), mk_v(:), mk_n(:), mkpe(:), mkpi(:), fcor(:), h_th(:), h_to(:),           &
    nudg(:,:), fnud(:,:,:), hdot(:,:), tide(:,:,:,:), bodf(:,:), taus(:,:),           &
    hlay(:,:), u(:,:), v(:,:), h_u(:,:), h_v(:,:), rs_h(:,:,:), dmdx(:,:,:),          &
    dmdy(:,:,:), v_cc(:,:), v_ll(:,:), tt3d(:,:,:), tb3d(:,:,:), tu3d(:,:,:)
  real(c_double), allocatable, target :: lid_p(:), lid_w(:), lid_s(:), lid_inv(:)   ! rgld = 1: pi_s, Ow, Os, 1/Osum (:64-67, 91)
  real(r4), allocatable, target :: h0r4(:,:)      ! what h_0.bin holds (ndeg, nlay)
  real(r4), allocatable, target :: rec_eta(:,:), rec_u(:,:), rec_v(:,:)   ! records formed on the GPU
  real(c_double), allocatable, target :: mnmx(:,:) ! (6, nlay): min/max of h, u, v per layer
  logical  :: h0_on_gpu = .false.
  integer,  allocatable :: posc(:)
  integer(c_int32_t), allocatable, target :: segm(:,:)   ! nudged open-boundary segments (nseg, 18), :1060-1240
  real(rw) :: w_ti(1), invf, ctim
  real(r8) :: tres
  logical  :: has_hdot, has_tide, has_bodf, nudging_on
  type(c_ptr) :: gpu = c_null_ptr
  type(c_ptr) :: gpus = c_null_ptr          ! BEOM_NGPU > 1: row bands on several devices (beom_multi_*)
  integer     :: ngpu = 1
  integer  :: out_rec = 0
  logical  :: out_ready = .false.

contains

! ------------------------------------------------------------------------------------
subroutine run()
  call setup_state()
  call advance()
  if ( c_associated(gpus) ) then
    if ( beom_multi_destroy(gpus) /= 0 ) continue
    gpus = c_null_ptr
  end if
  if ( c_associated(gpu) ) then
    if ( beom_destroy(gpu) /= 0 ) continue
  end if
end subroutine run

subroutine fail(code, text)                       ! errc/errm convention, shared_mod.f95:113-157
  integer, intent(in)      :: code
  character(*), intent(in) :: text
  errc = code
  errm = trim(errm) // ' ' // text
  call quit()
end subroutine fail

subroutine gpu_check(rc, cmsg, where)
  integer(c_int), intent(in)           :: rc
  character(kind = c_char), intent(in) :: cmsg(:)
  character(*), intent(in)             :: where
  if ( rc /= 0 ) call fail( int(rc), 'in ' // where // ' (libbeom_hip): ' // trim(c_to_f(cmsg)) )
end subroutine gpu_check

! ------------------------------------------------------------------------------------
! Initialisation = read_input_data (:105-250) and what it calls.
subroutine setup_state()
  real(rw), allocatable :: h_2d(:,:)
  real(r8), allocatable :: h_0(:,:)
  real(r8) :: dmin, dmax
  integer  :: ilay, lerm

  lerm = len_trim(errm)
  errm = trim(errm) // ' in subroutine setup_state of the MI355X host (beom_host_mod.f95),'
  call allocate_defaults()
  call check_options()

  allocate( h_2d(-1:lm+2, -1:mm+2), h_0(0:ndeg, nlay) )
  h_0 = 0._r8
  h_2d = 0._rw
  h_2d(1:lm, 1:mm) = cext**2._rw / grav                          ! default flat depth (:121)
  call load_depth( h_2d )
  call pack_cells( h_2d )

  dmin = minval( real(h_2d, r8), mask = h_2d > hdry )
  dmax = maxval( real(h_2d, r8) )
  if ( ocrp < 0.5_rw .and. nlay > 1 ) then                       ! (:137-152)
    if ( (real(topl(nlay), r8) * dmax + 10._r8 * real(hmin, r8)) >= dmin ) &
      call fail( -1, 'Please modify topl so that bathymetry is contained within lower layer.' )
  else if ( ocrp < 0.5_rw ) then
    if ( dmin <= 10._r8 * real(hmin, r8) ) &
      call fail( -1, 'Please adjust h_bo or hmin so that min(h_bo) > 10. * hmin.' )
  end if

  if ( ocrp < 0.5_rw ) then
    call rest_thickness_layered( h_2d, dmax, h_0 )
  else
    write(ioso, *) 'Calculating equilibrium thickness h_0 of layers...'
    call rest_thickness_outcrop( h_2d, h_0 )
    write(ioso, *) 'Completed the calculation of h_0.'
  end if

  if ( rgld > 0.5_rw ) call lid_operators( h_0 )

  allocate( h0r4(ndeg, nlay) )
  h0r4 = real( h_0(1:, :), r4 )
  call put_record_r4( 'h_0.bin', 1, h0r4, fresh = .true. )

  do ilay = 1, nlay
    hlay(:, ilay) = h_0(:, ilay) * real( mk_n(:), rw )           ! start from rest (:198-200)
  end do

  call load_nudging( h_2d )
  call load_initial()
  call load_small_forcings()
  call load_coriolis()

  invf = sum( fcor(:) ) / real( size(fcor(:)), rw )              ! (:223-229)
  if ( abs(invf) > 1.25e-5_rw ) then
    invf = 1._rw / invf
  else
    invf = 0._rw
  end if

  if ( rsta < 0.5_rw ) call write_parameter_echo()
  write(ioso, *) 'lm = ', lm
  write(ioso, *) 'mm = ', mm

  if ( rsta > 0.5_rw ) then
    call resume_from_outputs()
  else
    call write_outputs()
  end if

  call gpu_start()
  deallocate( h_2d, h_0 )
  if ( errc /= 0 ) call quit()
  errm = errm(1:lerm)
end subroutine setup_state

! ---- rigid lid: start pressure and the west / south / inverse-sum operators of its Poisson equation (:505-563) ----
subroutine lid_operators(h_0)
  real(r8), intent(in) :: h_0(0:, :)
  real(rw), allocatable :: total(:)
  real(rw) :: column
  integer  :: n, k, i, j
  logical  :: in_x, in_y
  allocate( lid_p(0:ndeg), lid_w(0:ndeg), lid_s(0:ndeg), lid_inv(0:ndeg), total(0:ndeg) )
  lid_w = 0._rw;  lid_s = 0._rw;  lid_inv = 0._rw;  total = 0._rw
  do n = 0, ndeg                                                 ! rest column minus depth, as a pressure
    column = h_0(n, 1)
    do k = 2, nlay
      column = column + h_0(n, k)
    end do
    lid_p(n) = ( column - h_th(n) ) * grav
  end do
  do n = 1, ndeg                                                 ! faces inside the frame carry the mean depth / dl**2
    i = subc(n, 1);  j = subc(n, 2)
    in_x = 1 < i .and. i < lm + 1
    in_y = 1 < j .and. j < mm + 1
    if ( in_x .and. (in_y .or. j == 1) ) lid_w(n) = 0.5_rw * ( h_th(n) + h_th(neig(5, n)) ) / dl**2
    if ( in_y .and. (in_x .or. i == 1) ) lid_s(n) = 0.5_rw * ( h_th(n) + h_th(neig(7, n)) ) / dl**2
  end do
  do n = 1, ndeg
    i = subc(n, 1);  j = subc(n, 2)
    if ( i < lm .and. j < mm ) then
      total(n) = lid_w(n) + lid_w(neig(1, n)) + lid_s(n) + lid_s(neig(3, n))
    else if ( i == lm .and. j < mm ) then
      total(n) = lid_w(n) + lid_s(n) + lid_s(neig(3, n))
    else if ( j == mm .and. i < lm ) then
      total(n) = lid_w(n) + lid_s(n) + lid_w(neig(1, n))
    else
      total(n) = lid_w(n) + lid_s(n)
    end if
    if ( i > 0 .and. i < lm + 1 .and. j > 0 .and. j < mm + 1 ) lid_inv(n) = 1 / total(n)
  end do
  deallocate( total )
end subroutine lid_operators

subroutine allocate_defaults()                                   ! initialize_variables (:252-307)
  allocate( neig(8, 0:ndeg), subc(0:ndeg, 2), posc(ndeg) )
  allocate( mk_u(0:ndeg), mk_v(0:ndeg), mk_n(0:ndeg), mkpe(0:ndeg), mkpi(0:ndeg) )
  allocate( fcor(0:ndeg), h_th(0:ndeg), h_to(0:ndeg) )
  allocate( nudg(0:ndeg, 3), fnud(0:ndeg, nlay, 3), hdot(0:ndeg, nlay) )
  allocate( tide(2, 1, 0:ndeg, 3), bodf(nlay, 2), taus(0:ndeg, 2) )
  allocate( hlay(0:ndeg, nlay), u(0:ndeg, nlay), v(0:ndeg, nlay) )
  allocate( h_u(0:ndeg, nlay), h_v(0:ndeg, nlay) )
  allocate( rs_h(2, 0:ndeg, nlay), dmdx(3, 0:ndeg, nlay), dmdy(3, 0:ndeg, nlay) )
  allocate( v_cc(0:ndeg, nlay), v_ll(0:ndeg, nlay) )
  allocate( tt3d(0:ndeg, 2, nlay), tb3d(0:ndeg, 2, nlay), tu3d(0:ndeg, 2, nlay) )
  neig = 0; subc = 0; posc = 0
  mk_u = 0._rw; mk_v = 0._rw; mk_n = 0._rw; mkpe = 0._rw; mkpi = 0._rw
  h_th = 0._rw; h_to = 0._rw
  nudg = 0._rw; fnud = 0._rw; hdot = 0._rw; tide = 0._rw; bodf = 0._rw
  hlay = 0._r8; u = 0._rw; v = 0._rw; h_u = 0._rw; h_v = 0._rw
  rs_h = 0._rw; dmdx = 0._rw; dmdy = 0._rw
  v_cc = bvis; v_ll = bvis
  tt3d = 0._rw; tb3d = 0._rw; tu3d = 0._rw
  w_ti = 0._rw
  fcor = f0
  taus(:, 1) = real( tauw )
  taus(:, 2) = aimag( tauw )
  ctim = 0._rw
  tres = 0._r8
  has_hdot = .false.; has_tide = .false.; has_bodf = .false.; nudging_on = .false.
end subroutine allocate_defaults

subroutine check_options()                                       ! check_consistency_options (:969-1058)
  logical :: there
  integer :: ios
  idir = adjustl(idir)
  odir = adjustl(odir)
  inquire( file = trim(idir), exist = there, iostat = ios )
  if ( .not. there ) call fail( -1, 'idir is set to ' // trim(idir) // &
                                ' but this directory does not exist. Program stopped.' )
  inquire( file = trim(odir), exist = there, iostat = ios )
  if ( .not. there ) call fail( -1, 'odir is set to ' // trim(odir) // &
                                ' but this directory does not exist. Program stopped.' )
  if ( idir(len_trim(idir):len_trim(idir)) /= '/' ) idir(len_trim(idir)+1:len_trim(idir)+1) = '/'
  if ( odir(len_trim(odir):len_trim(odir)) /= '/' ) odir(len_trim(odir)+1:len_trim(odir)+1) = '/'
  if ( lm < 1 .or. mm < 1 ) call fail( -1, 'grid dimensions (lm,mm) should be >= 1.' )
  if ( dl < 1.e1_rw ) call fail( -1, 'mesh size (dl) should be >= 10 meters.' )
  if ( abs(f0) > 2.e-4_rw ) call fail( -1, 'Coriolis parameter (f0, in s**(-1)) should be within: ' // &
                                       '-2x10**(-4) < f0 < 2x10**(-4).' )
  if ( dvis < 0._rw .or. dvis > 5._rw ) call fail( -1, 'Viscosity coefficient should be within: 0 <= dvis < 5.0.' )
  if ( (bdrg < 0._rw .or. bdrg > 15.e-3_rw) .and. qdrg > 0.5_rw ) then
    call fail( -1, 'quadratic bottom drag coefficient bdrg should be within: 0 <= bdrg < 5x10**(-3).' )
  else if ( bdrg < 0._rw .or. bdrg > 5.e-2_rw ) then
    call fail( -1, 'linear bottom drag coefficient bdrg has units of m s**(-1) and should be within: ' // &
                   '0 <= bdrg < 5x10**(-3) x u_max.' )
  end if
#ifdef BEOM_FORK_SWITCHES
  if ( (tdrg < 0._rw .or. tdrg > 15.e-3_rw) .and. qdrg > 0.5_rw ) then                    ! :1043-1051
    call fail( -1, 'quadratic top drag coefficient tdrg should be within: 0 <= tdrg < 5x10**(-3).' )
  else if ( tdrg < 0._rw .or. tdrg > 5.e-2_rw ) then
    call fail( -1, 'linear top drag coefficient tdrg has units of m s**(-1) and should be within: ' // &
                   '0 <= tdrg < 5x10**(-3) x u_max.' )
  end if
#endif
  if ( rgld > 0.5_rw .and. ocrp < 0.5_rw ) &
    call fail( -5, 'rgld = 1 (rigid lid) needs ocrp = 1: the lid operators are only set up with outcropping.' )
  if ( nlay > beom_max_layers ) call fail( -3, 'nlay exceeds BEOM_MAX_LAYERS of libbeom_hip.' )
end subroutine check_options

! ---- files of the testcases/*.m contract: real*4, little-endian, direct access ------
logical function input_exists(keyw)
  character(4), intent(in) :: keyw
  integer :: ios
  inquire( file = trim(idir) // keyw // '.bin', exist = input_exists, iostat = ios )
end function input_exists

subroutine get_input_r4(keyw, buf)                               ! read_input_file I/O part (:803-826)
  character(4), intent(in) :: keyw
  real(r4), intent(out)    :: buf(:)
  integer :: lrec, unum, ios
  inquire( iolength = lrec ) buf
  unum = get_un()
  open( unit = unum, file = trim(idir) // keyw // '.bin', status = 'old', action = 'read', &
        access = 'direct', form = 'unformatted', recl = lrec, iostat = ios )
  if ( ios == 0 ) read( unum, rec = 1, iostat = ios ) buf
  if ( ios /= 0 ) call fail( ios, 'could not open/read file ' // keyw // '.bin from directory ' // trim(idir) )
  close( unum )
end subroutine get_input_r4

subroutine put_record_r4(fname, irec, buf, fresh)
  character(*), intent(in) :: fname
  integer, intent(in)      :: irec
  real(r4), intent(in)     :: buf(:,:)
  logical, intent(in)      :: fresh
  integer :: lrec, unum, ios
  inquire( iolength = lrec ) buf
  unum = get_un()
  if ( fresh ) then
    open( unit = unum, file = trim(odir) // fname, status = 'replace', action = 'write', &
          access = 'direct', form = 'unformatted', recl = lrec, iostat = ios )
  else
    open( unit = unum, file = trim(odir) // fname, status = 'old', action = 'write', &
          access = 'direct', form = 'unformatted', recl = lrec, iostat = ios )
  end if
  if ( ios == 0 ) write( unum, rec = irec, iostat = ios ) buf
  close( unum )
  if ( ios /= 0 ) call fail( ios, 'could not write ' // fname // ' in ' // trim(odir) )
end subroutine put_record_r4

subroutine load_depth(h_2d)                                      ! keyw 'h_bo' (:827-839)
  real(rw), intent(inout) :: h_2d(-1:, -1:)
  real(r4), allocatable   :: a(:), top(:)
  if ( .not. input_exists('h_bo') ) return
  allocate( a((lm+2) * (mm+2)) )
  call get_input_r4( 'h_bo', a )
  if ( FORK_TOPT > 0.5_rw ) then                                 ! surface topography, subtracted in real*4 (:809-832)
    allocate( top((lm+2) * (mm+2)) )
    call get_input_r4( 'h_to', top )
    a = a - top
    deallocate( top )
  end if
  h_2d = 0._rw
  h_2d(0:lm+1, 0:mm+1) = real( reshape(a, (/ lm+2, mm+2 /)), rw )
  where ( h_2d < hdry ) h_2d = 0._rw
  h_2d(0, :) = 0._rw;  h_2d(:, 0) = 0._rw;  h_2d(lm+1, :) = 0._rw;  h_2d(:, mm+1) = 0._rw
  deallocate( a )
end subroutine load_depth

! ---- index_grid_points (:567-764): packed numbering, neighbours, masks, grid.bin ----
subroutine pack_cells(h_2d)
  real(rw), intent(in)  :: h_2d(-1:, -1:)
  integer, allocatable  :: look(:,:)
  integer(i4), allocatable :: rec(:)
  logical, allocatable  :: wet(:,:)
  character(sstr) :: txt
  integer :: i, j, n, lrec, unum

  allocate( look(-1:lm+2, -1:mm+2), wet(-1:lm+2, -1:mm+2) )
  wet  = h_2d > hdry
  look = 0
  n = 0
  do j = 0, mm + 1
    do i = 0, lm + 1
      if ( wet(i,j) .or. wet(i-1,j) .or. wet(i,j-1) .or. wet(i-1,j-1) ) then
        n = n + 1
        look(i,j) = n
      end if
    end do
  end do
  if ( n /= ndeg ) then
    write(txt, '(1i8)') n
    call fail( min(-1, -n), 'wrong input parameter! Please set ndeg = ' // trim(txt) // &
                            ' inside file shared_mod.f95.' )
  end if

  if ( xper > 0.5_rw ) then                                      ! (:614-640)
    do j = 1, mm
      if ( wet(1,j) .and. wet(lm,j) ) then
        look(0,j) = look(lm,j);  look(lm+1,j) = look(1,j)
        mk_u( look(1,j) ) = 1._rw
      end if
      if ( j > 1 ) then
        if ( wet(1,j-1) .and. wet(1,j) .and. wet(lm,j-1) .and. wet(lm,j) ) mkpe( look(1,j) ) = 1._rw
      end if
      if ( j == mm .and. wet(1,j) .and. wet(lm,j) ) then
        look(0,mm+1) = look(lm,mm+1);  look(lm+1,mm+1) = look(1,mm+1)
      end if
    end do
  end if
  if ( yper > 0.5_rw ) then                                      ! (:642-668)
    do i = 1, lm
      if ( wet(i,1) .and. wet(i,mm) ) then
        look(i,0) = look(i,mm);  look(i,mm+1) = look(i,1)
        mk_v( look(i,1) ) = 1._rw
      end if
      if ( i > 1 ) then
        if ( wet(i-1,1) .and. wet(i,1) .and. wet(i-1,mm) .and. wet(i,mm) ) mkpe( look(i,1) ) = 1._rw
      end if
      if ( i == lm .and. wet(i,mm) .and. wet(i,1) ) then
        look(lm+1,0) = look(lm+1,mm);  look(lm+1,mm+1) = look(lm+1,1)
      end if
    end do
  end if
  if ( xper > 0.5_rw .and. yper > 0.5_rw ) then                  ! (:672-685)
    if ( wet(1,1) .and. wet(lm,1) .and. wet(1,mm) ) then
      look(0,0) = look(lm,mm);  mkpe( look(1,1) ) = 1._rw;  look(0,mm+1) = look(lm,1)
    end if
    if ( wet(lm,mm) .and. wet(1,mm) .and. wet(lm,1) ) then
      look(lm+1,0) = look(1,mm);  look(lm+1,mm+1) = look(1,1)
    end if
  end if

  n = 0
  do j = 0, mm + 1
    do i = 0, lm + 1
      if ( .not. ( wet(i,j) .or. wet(i-1,j) .or. wet(i,j-1) .or. wet(i-1,j-1) ) ) cycle
      n = n + 1
      if ( wet(i,j) )                                              mk_n(n) = 1._rw
      if ( wet(i,j) .and. wet(i-1,j) )                             mk_u(n) = 1._rw
      if ( wet(i,j) .and. wet(i,j-1) )                             mk_v(n) = 1._rw
      if ( wet(i,j) .and. wet(i-1,j) .and. wet(i,j-1) .and. wet(i-1,j-1) ) mkpe(n) = 1._rw
      mkpi(n) = 1._rw
      posc(n) = i + 1 + j * (lm + 2)
      subc(n, 1) = i;  subc(n, 2) = j
      neig(1, n) = look(i+1, j  );  neig(2, n) = look(i+1, j+1)
      neig(3, n) = look(i,   j+1);  neig(4, n) = look(i-1, j+1)
      neig(5, n) = look(i-1, j  );  neig(6, n) = look(i-1, j-1)
      neig(7, n) = look(i,   j-1);  neig(8, n) = look(i+1, j-1)
      h_th(n) = h_2d(i, j)
    end do
  end do
  h_th(0) = h_2d(0, 0)

  allocate( rec(ndeg) )                                          ! grid.bin: 5 int32 records (:732-748)
  inquire( iolength = lrec ) rec
  unum = get_un()
  open( unit = unum, file = trim(odir) // 'grid.bin', status = 'replace', action = 'write', &
        access = 'direct', form = 'unformatted', recl = lrec )
  rec = int( posc, i4 );          write( unum, rec = 1 ) rec
  rec = nint( mk_n(1:), i4 );     write( unum, rec = 2 ) rec
  rec = nint( mk_u(1:), i4 );     write( unum, rec = 3 ) rec
  rec = nint( mk_v(1:), i4 );     write( unum, rec = 4 ) rec
  rec = nint( mkpi(1:), i4 );     write( unum, rec = 5 ) rec
  close( unum )
  deallocate( rec, look, wet )
end subroutine pack_cells

! ---- resting thickness without outcrops (:154-175) ---------------------------------
subroutine rest_thickness_layered(h_2d, dmax, h_0)
  real(rw), intent(in)    :: h_2d(-1:, -1:)
  real(r8), intent(in)    :: dmax
  real(r8), intent(inout) :: h_0(0:, :)
  real(r8) :: above, below
  integer  :: n, k, m
  do n = 1, ndeg
    if ( mk_n(n) < 0.5_rw ) cycle
    do k = nlay, 1, -1
      above = 0._r8
      below = 0._r8
      if ( k > 1 ) above = dmax * real( topl(k), r8 )
      do m = k + 1, nlay
        below = below + h_0(n, m)
      end do
      h_0(n, k) = real( h_2d(subc(n,1), subc(n,2)), r8 ) - above - below
    end do
  end do
end subroutine rest_thickness_layered

! ---- resting thickness with Salmon's outcrop term (:309-502): per-column Newton
!      iteration on the hydrostatic balance with over-relaxation.  The elimination step
!      re-reads its multiplier inside the column loop exactly as the reference does
!      (:446-454); changing that would change h_0.bin in the last digits. ----------
subroutine rest_thickness_outcrop(h_2d, h_0)
  real(rw), intent(in)    :: h_2d(-1:, -1:)
  real(r8), intent(inout) :: h_0(0:, :)
  real(r8) :: rho(nlay), deep(nlay), cons(nlay), g(nlay), f(nlay), a(nlay, nlay+1), row(nlay+1)
  real(r8) :: dmax, hbot, tol, hs, s, big, acc
  integer  :: n, k, m, l, it, piv
  logical  :: done
  character(sstr) :: txt

  tol = real( tole, r8 )
  write(ioso, *) 'Tolerance = ', tole, ' meters.'
  dmax = real( maxval(h_2d), r8 )
  rho  = real( rhon(:), r8 )
  hs   = real( hsal, r8 )
  do k = 1, nlay                                                 ! layer thickness at the deepest point
    deep(k) = dmax * ( 1._r8 - real(topl(k), r8) )
    if ( k < nlay ) deep(k) = deep(k) - dmax * ( 1._r8 - real(topl(k+1), r8) )
  end do
  do k = 1, nlay
    s = 0._r8
    do m = 1, nlay
      s = s + deep(m)
    end do
    cons(k) = dmax * (-1._r8) + s
    do m = 1, k - 1
      cons(k) = cons(k) - ( rho(k) - rho(m) ) * deep(m) / rho(k)
    end do
  end do

  do n = 1, ndeg
    if ( mk_n(n) < 0.5_rw ) cycle
    hbot = real( h_2d(subc(n,1), subc(n,2)), r8 )
    do k = nlay, 1, -1                                           ! first guess
      s = 0._r8
      do m = k + 1, nlay
        s = s + g(m)
      end do
      g(k) = max( hbot - dmax * real(topl(k), r8) - s, hs )
    end do
    done = .false.
    do it = 1, itmx
      s = 0._r8
      do m = 1, nlay
        s = s + g(m)
      end do
      do k = 1, nlay                                             ! residual of the balance
        f(k) = ( hbot - s ) + 1._r8 / real(nsal - 1, r8) * hs * (hs / g(k))**(nsal - 1) + cons(k)
        f(k) = f(k) * (-1._r8)
        do m = 1, k - 1
          f(k) = f(k) - ( rho(k) - rho(m) ) * g(m) / rho(k)
        end do
      end do
      if ( it == itmx ) exit
      if ( all( abs(f) < tol ) ) then
        h_0(n, :) = g
        done = .true.
        exit
      end if
      do k = 1, nlay                                             ! Jacobian | rhs
        do m = 1, nlay
          a(k, m) = min( rho(k), rho(m) ) / rho(k)
          if ( k == m ) a(k, m) = a(k, m) + (hs / g(m))**nsal
        end do
      end do
      a(:, nlay+1) = f * (-1._r8)
      do k = 1, nlay                                             ! elimination with row pivoting
        piv = 0
        big = 0._r8
        do m = k, nlay
          if ( abs(a(m, k)) > big ) then
            big = abs(a(m, k));  piv = m
          end if
        end do
        if ( piv /= k ) then
          row = a(k, :);  a(k, :) = a(piv, :);  a(piv, :) = row
        end if
        do m = k + 1, nlay
          do l = k, nlay + 1
            a(m, l) = a(m, l) - a(k, l) * ( a(m, k) / a(k, k) )
          end do
          a(m, k) = 0._r8
        end do
      end do
      do k = nlay, 1, -1                                         ! back substitution
        acc = 0._r8
        do m = k + 1, nlay
          acc = acc + a(k, m) * a(m, nlay+1)
        end do
        a(k, nlay+1) = ( a(k, nlay+1) - acc ) / a(k, k)
      end do
      g = (1._r8 - sor) * g + sor * ( a(:, nlay+1) + g )
      if ( any( g <= tol ) ) g = max( g, tol )
    end do
    if ( .not. done ) then
      write(txt, *) hbot
      call fail( -n, 'calculation of h_layers did not converge, local depth (meters) is ' // txt )
    end if
  end do
end subroutine rest_thickness_outcrop

! ---- nudg.bin (:843-881) --------------------------------------------------------------
subroutine load_nudging(h_2d)
  real(rw), intent(in)  :: h_2d(-1:, -1:)
  real(r4), allocatable :: a(:), c(:,:,:)
  integer :: n, i, j, k, nseg
  if ( .not. input_exists('nudg') ) return
  allocate( a((lm+2) * (mm+2) * 3), c(0:lm+1, 0:mm+1, 3) )
  call get_input_r4( 'nudg', a )
  c = reshape( a, (/ lm+2, mm+2, 3 /) )
  do n = 1, ndeg
    i = subc(n, 1);  j = subc(n, 2)
    nudg(n, ix_n) = real( c(i, j, ix_n), rw )
    nudg(n, ix_u) = 0._rw
    nudg(n, ix_v) = 0._rw
    if ( i >= 1 ) then
      if ( c(i-1, j, ix_u) > 1.e-9_r4 .and. c(i, j, ix_u) > 1.e-9_r4 ) &
        nudg(n, ix_u) = real( c(i, j, ix_u), rw ) * 0.5_rw + real( c(i-1, j, ix_u), rw ) * 0.5_rw
    end if
    if ( j >= 1 ) then
      if ( c(i, j-1, ix_v) > 1.e-9_r4 .and. c(i, j, ix_v) > 1.e-9_r4 ) &
        nudg(n, ix_v) = real( c(i, j, ix_v), rw ) * 0.5_rw + real( c(i, j-1, ix_v), rw ) * 0.5_rw
    end if
  end do
  if ( any( nudg > 1.e-9_rw ) ) then
    nudging_on = .true.
    call find_open_boundaries( h_2d, c )
  end if
  do k = 1, nlay
    fnud(1:, k, ix_n) = real( hlay(1:, k), rw )
  end do
  deallocate( a, c )
end subroutine load_nudging

! ---- index_boundary_points (:1060-1240): one row per nudged open-boundary segment; the 18
!      columns are those of the reference (1-3 normal velocity point, 4/5 zonal/meridional
!      flag, 6 orientation, 7-9 dry cell, 10-12 wet cell, 13-15 interior normal point,
!      16-18 interior cell).  Two passes: count, then fill. ----------------------------------
subroutine find_open_boundaries(h_2d, c)
  real(rw), intent(in) :: h_2d(-1:, -1:)
  real(r4), intent(in) :: c(0:, 0:, :)
  integer, allocatable :: look(:,:)
  integer :: i, j, n, pass, nseg
  logical :: wet, wetw, wets
  allocate( look(-1:lm+2, -1:mm+2) )
  look = 0
  n = 0
  do j = 0, mm + 1
    do i = 0, lm + 1
      if ( h_2d(i,j) > hdry .or. h_2d(i-1,j) > hdry .or. h_2d(i,j-1) > hdry .or. h_2d(i-1,j-1) > hdry ) then
        n = n + 1
        look(i,j) = n
      end if
    end do
  end do
  do pass = 1, 2
    nseg = 0
    do j = 0, mm + 1
      do i = 0, lm + 1
        wet = h_2d(i,j) > hdry;  wetw = h_2d(i-1,j) > hdry;  wets = h_2d(i,j-1) > hdry
        if ( wet .and. .not. wetw .and. xper < 0.5_rw ) then                      ! western boundary
          if ( c(i,j,ix_u) > tiny(0._r4) .and. c(max(i-1,0),j,ix_u) > tiny(0._r4) ) then
            nseg = nseg + 1
            if ( pass == 2 ) segm(nseg, :) = (/ look(i,j), i, j, 1, 0, 1, look(i-1,j), i-1, j, look(i,j), i, j, &
                                                look(i+1,j), i+1, j, look(i+1,j), i+1, j /)
          end if
        end if
        if ( .not. wet .and. wetw .and. xper < 0.5_rw ) then                      ! eastern boundary
          if ( c(max(i-1,0),j,ix_u) > tiny(0._r4) .and. c(i,j,ix_u) > tiny(0._r4) ) then
            nseg = nseg + 1
            if ( pass == 2 ) segm(nseg, :) = (/ look(i,j), i, j, 1, 0, -1, look(i,j), i, j, look(i-1,j), i-1, j, &
                                                look(i-1,j), i-1, j, look(i-2,j), i-2, j /)
          end if
        end if
        if ( wet .and. .not. wets .and. yper < 0.5_rw ) then                      ! southern boundary
          if ( c(i,j,ix_v) > tiny(0._r4) .and. c(i,max(j-1,0),ix_v) > tiny(0._r4) ) then
            nseg = nseg + 1
            if ( pass == 2 ) segm(nseg, :) = (/ look(i,j), i, j, 0, 1, 1, look(i,j-1), i, j-1, look(i,j), i, j, &
                                                look(i,j+1), i, j+1, look(i,j+1), i, j+1 /)
          end if
        end if
        if ( .not. wet .and. wets .and. yper < 0.5_rw ) then                      ! northern boundary
          if ( c(i,max(j-1,0),ix_v) > tiny(0._r4) .and. c(i,j,ix_v) > tiny(0._r4) ) then
            nseg = nseg + 1
            if ( pass == 2 ) segm(nseg, :) = (/ look(i,j), i, j, 0, 1, -1, look(i,j), i, j, look(i,j-1), i, j-1, &
                                                look(i,j-1), i, j-1, look(i,j-2), i, j-2 /)
          end if
        end if
      end do
    end do
    if ( nseg == 0 ) call fail( -1, 'the nudged open boundary segments could not be identified.' )
    if ( pass == 1 ) allocate( segm(nseg, 18) )
  end do
  deallocate( look )
end subroutine find_open_boundaries

! ---- init.bin (:882-910): interface anomalies -> thickness, velocities ----------------
subroutine load_initial()
  real(r4), allocatable :: a(:), c(:,:,:,:)
  integer :: n, i, j, k
  if ( .not. input_exists('init') ) return
  allocate( a((lm+2) * (mm+2) * nlay * 3), c(0:lm+1, 0:mm+1, nlay, 3) )
  call get_input_r4( 'init', a )
  c = reshape( a, (/ lm+2, mm+2, nlay, 3 /) )
  do k = 1, nlay
    do n = 1, ndeg
      i = subc(n, 1);  j = subc(n, 2)
      if ( k < nlay ) then
        fnud(n, k, ix_n) = real( hlay(n, k) + real(c(i,j,k,ix_n), r8) - real(c(i,j,k+1,ix_n), r8), rw )
      else
        fnud(n, k, ix_n) = real( hlay(n, k) + real(c(i,j,k,ix_n), r8), rw )
      end if
      fnud(n, k, ix_n) = fnud(n, k, ix_n) * mk_n(n)
      fnud(n, k, ix_u) = real( c(i,j,k,ix_u), rw )
      fnud(n, k, ix_v) = real( c(i,j,k,ix_v), rw )
      if ( rsta < 0.5_rw ) then
        hlay(n, k) = real( fnud(n, k, ix_n) * mk_n(n), r8 )
        u(n, k) = fnud(n, k, ix_u)
        v(n, k) = fnud(n, k, ix_v)
      end if
    end do
  end do
  deallocate( a, c )
end subroutine load_initial

! ---- bodf.bin, hdot.bin, taus.bin, tide.bin (:840-842, 911-931, 951-964) ------------
subroutine load_small_forcings()
  real(r4), allocatable :: a(:), c3(:,:,:), c5(:,:,:,:,:)
  integer :: n, k
  if ( input_exists('bodf') ) then
    allocate( a(nlay * 2) )
    call get_input_r4( 'bodf', a )
    bodf = real( reshape(a, (/ nlay, 2 /)), rw )
    has_bodf = .true.
    deallocate( a )
  end if
  if ( input_exists('hdot') ) then
    allocate( a((lm+2) * (mm+2) * nlay), c3(0:lm+1, 0:mm+1, nlay) )
    call get_input_r4( 'hdot', a )
    c3 = reshape( a, (/ lm+2, mm+2, nlay /) )
    do k = 1, nlay
      do n = 1, ndeg
        hdot(n, k) = real( c3(subc(n,1), subc(n,2), k), rw )
      end do
    end do
    has_hdot = .true.
    deallocate( a, c3 )
  end if
  if ( input_exists('taus') ) then
    allocate( a((lm+2) * (mm+2) * 2), c3(0:lm+1, 0:mm+1, 2) )
    call get_input_r4( 'taus', a )
    c3 = reshape( a, (/ lm+2, mm+2, 2 /) )
    taus = 0._rw
    do n = 1, ndeg
      taus(n, 1) = real( c3(subc(n,1), subc(n,2), 1), rw )
      taus(n, 2) = real( c3(subc(n,1), subc(n,2), 2), rw )
    end do
    deallocate( a, c3 )
  end if
  if ( input_exists('tide') ) then
    allocate( a(2 * size(w_ti) * (lm+2) * (mm+2) * 3), c5(2, size(w_ti), 0:lm+1, 0:mm+1, 3) )
    call get_input_r4( 'tide', a )
    c5 = reshape( a, (/ 2, size(w_ti), lm+2, mm+2, 3 /) )
    do k = 1, size(w_ti)
      w_ti(k) = real( c5(1, k, 0, 0, 1), rw )
      write(ioso, *) 'Tidal constituent (rad/day) = ', w_ti(k)
    end do
    do n = 1, ndeg
      tide(:, :, n, :) = real( c5(:, :, subc(n,1), subc(n,2), :), rw )
    end do
    has_tide = .true.
    deallocate( a, c5 )
  end if
end subroutine load_small_forcings

! ---- fcor.bin (:932-950): psi-point average IN real*4, domain mean at the sentinel --
subroutine load_coriolis()
  real(r4), allocatable :: a(:), c(:,:)
  integer :: n, i, j
  if ( .not. input_exists('fcor') ) return
  allocate( a((lm+2) * (mm+2)), c(0:lm+1, 0:mm+1) )
  call get_input_r4( 'fcor', a )
  c = reshape( a, (/ lm+2, mm+2 /) )
  fcor(0) = real( sum(c) / real(size(c), r4), rw )
  do n = 1, ndeg
    i = subc(n, 1);  j = subc(n, 2)
    if ( i > 0 .and. j > 0 ) then
      fcor(n) = real( c(i,j) * 0.25_r4 + c(i-1,j) * 0.25_r4 + c(i,j-1) * 0.25_r4 + c(i-1,j-1) * 0.25_r4, rw )
    else
      fcor(n) = real( c(i,j), rw )
    end if
  end do
  deallocate( a, c )
end subroutine load_coriolis

! ---- param_basin.txt (:1242-1297): one Octave-evaluable assignment per line ---------
subroutine write_parameter_echo()
  integer  :: unum, ios
  real(rw) :: zero
  zero = 0._rw
  unum = get_un()
  open( unit = unum, file = trim(odir) // 'param_basin.txt', action = 'write', status = 'replace', iostat = ios )
  write(unum, *) 'lm             = ',  lm,              ';'
  write(unum, *) 'mm             = ',  mm,              ';'
  write(unum, *) 'nlay           = ',  nlay,            ';'
  write(unum, *) 'ndeg           = ',  ndeg,            ';'
  write(unum, *) 'dl             = ',  dl,              ';'
  write(unum, *) 'cext           = ',  cext,            ';'
  write(unum, *) 'f0             = ',  f0,              ';'
  write(unum, *) 'rhon           = [', rhon(1:nlay),   '];'
  write(unum, *) 'topl           = [', topl(1:nlay),   '];'
  write(unum, *) 'dt_s           = ',  dt_s,            ';'
  write(unum, *) 'dt_o           = ',  dt_o,            ';'
  write(unum, *) 'dt_r           = ',  dt_r,            ';'
  write(unum, *) 'dt3d           = ',  dt3d,            ';'
  write(unum, *) 'bvis           = ',  bvis,            ';'
  write(unum, *) 'dvis           = ',  dvis,            ';'
  zero = FORK_SVIS
  write(unum, *) 'svis           = ',  zero,            ';'
  write(unum, *) 'bdrg           = ',  bdrg,            ';'
  zero = FORK_TDRG
  write(unum, *) 'tdrg           = ',  zero,            ';'
  write(unum, *) 'tole           = ',  tole,            ';'
  write(unum, *) 'nsal           = ',  nsal,            ';'
  write(unum, *) 'hsal           = ',  hsal,            ';'
  write(unum, *) 'hmin           = ',  hmin,            ';'
  write(unum, *) 'hdry           = ',  hdry,            ';'
  write(unum, *) 'hsbl           = ',  hsbl,            ';'
  write(unum, *) 'hbbl           = ',  hsbl,            ';'
  write(unum, *) 'g_fb           = ',  g_fb,            ';'
  write(unum, *) 'uadv           = ',  uadv,            ';'
  write(unum, *) 'qdrg           = ',  qdrg,            ';'
  write(unum, *) 'ocrp           = ',  ocrp,            ';'
  write(unum, *) 'tauwx          = ',  real(tauw),      ';'
  write(unum, *) 'tauwy          = ',  aimag(tauw),     ';'
  write(unum, *) 'rsta           = ',  rsta,            ';'
  write(unum, *) 'xper           = ',  xper,            ';'
  write(unum, *) 'yper           = ',  yper,            ';'
  write(unum, *) 'diag           = ',  diag,            ';'
  write(unum, *) 'rgld           = ',  rgld,            ';'
  write(unum, *) 'mcbc           = ',  mcbc,            ';'
  zero = FORK_TOPT
  write(unum, *) 'topt           = ',  zero,            ';'
  write(unum, *) 'idir           = ', '''', trim(idir), '''', ';'
  write(unum, *) 'desc           = ', '''', trim(desc), '''', ';'
  write(unum, *) 'dt             = ',  dt,              ';'
  close( unum )
end subroutine write_parameter_echo

! ---- restart (:1299-1420): last complete record of time.txt / u___ / v___ / eta_ ----
subroutine resume_from_outputs()
  real(r4), allocatable :: a(:,:)
  real(r8) :: stamp
  integer  :: unum, ios, nrec, lrec, k, n
  unum = get_un()
  nrec = 0
  open( unit = unum, file = trim(odir) // 'time.txt', form = 'formatted', action = 'read', status = 'old', iostat = ios )
  do while ( ios == 0 )
    read( unum, *, iostat = ios ) stamp
    if ( ios == 0 ) then
      nrec = nrec + 1;  tres = stamp
    end if
  end do
  close( unum )
  write(ioso, *) '*** Restarting from record number ', nrec, ' at time = ', real(tres, rw)
  allocate( a(ndeg, nlay) )
  inquire( iolength = lrec ) a
  unum = get_un()
  open( unit = unum, file = trim(odir) // 'u___.bin', access = 'direct', form = 'unformatted', recl = lrec, &
        status = 'old', action = 'read', iostat = ios )
  if ( ios == 0 ) read( unum, rec = nrec, iostat = ios ) a
  close( unum )
  if ( ios /= 0 ) call fail( ios, 'could not open/read file u___.bin from directory ' // trim(odir) )
  u(1:ndeg, :) = real( a, rw )
  open( unit = unum, file = trim(odir) // 'v___.bin', access = 'direct', form = 'unformatted', recl = lrec, &
        status = 'old', action = 'read', iostat = ios )
  if ( ios == 0 ) read( unum, rec = nrec, iostat = ios ) a
  close( unum )
  if ( ios /= 0 ) call fail( ios, 'could not open/read file v___.bin from directory ' // trim(odir) )
  v(1:ndeg, :) = real( a, rw )
  open( unit = unum, file = trim(odir) // 'eta_.bin', access = 'direct', form = 'unformatted', recl = lrec, &
        status = 'old', action = 'read', iostat = ios )
  if ( ios == 0 ) read( unum, rec = nrec, iostat = ios ) a
  close( unum )
  if ( ios /= 0 ) call fail( ios, 'could not open/read file eta_.bin from directory ' // trim(odir) )
  do k = 1, nlay                                                 ! interface elevation -> thickness
    do n = 1, ndeg
      if ( k < nlay ) then
        hlay(n, k) = real(h0r4(n, k), r8) + real(a(n, k), r8) - real(a(n, k+1), r8)
      else
        hlay(n, k) = real(h0r4(n, k), r8) + real(a(n, k), r8)
      end if
      hlay(n, k) = hlay(n, k) * real( mk_n(n), r8 )
    end do
  end do
  deallocate( a )
  out_ready = .true.
  out_rec   = nrec + 1
end subroutine resume_from_outputs

! ---- outputs (:2681-3001) ----------------------------------------------------------
subroutine write_outputs()
  integer  :: unum, ios, k
  character(9) :: txt
  logical  :: fresh
  fresh = .not. out_ready
  if ( fresh ) out_rec = 1
  call write_field( 'eta_', fresh )
  call write_field( 'u___', fresh )
  call write_field( 'v___', fresh )
  if ( diag > 0.5_rw ) then
    call write_field( 'pvor', fresh )
    call write_field( 'mont', fresh )
    call write_field( 'v_cc', fresh )
  end if
  unum = get_un()                                                ! commit point: time.txt last (:2724-2738)
  if ( fresh ) then
    open( unit = unum, file = trim(odir) // 'time.txt', form = 'formatted', action = 'write', status = 'replace', iostat = ios )
  else
    open( unit = unum, file = trim(odir) // 'time.txt', form = 'formatted', action = 'write', status = 'old', &
          position = 'append', iostat = ios )
  end if
  write( unum, * ) real(ctim, r8)
  close( unum )
  out_rec   = out_rec + 1
  out_ready = .true.

  write(ioso, *) 'ctim = ', ctim, ' days; dt_s = ', dt_s, ' days; record = ', out_rec - 1
  do k = 1, nlay
    write(ioso, *) 'min/max h', k, '= ', minval( real(hlay(:,k), rw), mask = mk_n > 0.5 ), &
                                         maxval( real(hlay(:,k), rw), mask = mk_n > 0.5 )
    if ( any( mk_u > 0.5 ) ) then
      write(ioso, *) 'min/max u', k, '= ', minval( u(:,k), mask = mk_u > 0.5 ), maxval( u(:,k), mask = mk_u > 0.5 )
    else
      write(ioso, *) 'min/max u', k, '= ', minval( u(:,k) ), maxval( u(:,k) )
    end if
    if ( any( mk_v > 0.5 ) ) then
      write(ioso, *) 'min/max v', k, '= ', minval( v(:,k), mask = mk_v > 0.5 ), maxval( v(:,k), mask = mk_v > 0.5 )
    else
      write(ioso, *) 'min/max v', k, '= ', minval( v(:,k) ), maxval( v(:,k) )
    end if
  end do
  do k = 1, nlay                                                 ! thin-layer abort (:2798-2808)
    if ( any( mk_n(1:) > 0.5_rw .and. hlay(1:, k) < real(0.5_rw * hmin, r8) ) ) then
      write(txt, '(1i8)') k
      call fail( -k, 'layer number ' // trim(txt) // ' has its thickness < hmin; Calculation halted.' )
    end if
  end do
end subroutine write_outputs

! Output record prepared on the device (beom_download_outputs): eta inversion, real*4
! conversion, min/max and the thin-layer scan of write_outputs (:2769-2808) without moving
! the FP64 state to the host.
subroutine write_outputs_from_gpu()
  character(kind = c_char) :: cmsg(lstr + 1)
  integer(c_int) :: rc, thin
  integer  :: unum, ios, k
  character(9) :: txt
  type(c_ptr) :: p_h0
  if ( .not. allocated(rec_eta) ) allocate( rec_eta(ndeg, nlay), rec_u(ndeg, nlay), rec_v(ndeg, nlay), mnmx(6, nlay) )
  cmsg = c_null_char
  p_h0 = c_null_ptr
  if ( .not. h0_on_gpu ) p_h0 = c_loc(h0r4)
  thin = 0
  if ( ngpu > 1 ) then                                           ! every band forms its own rows on its device
    rc = beom_multi_download_outputs( gpus, c_loc(h0r4), c_loc(rec_eta), c_loc(rec_u), c_loc(rec_v), c_loc(mnmx), thin, &
                                      cmsg, int(lstr, c_int) )
  else
    rc = beom_download_outputs( gpu, p_h0, c_loc(rec_eta), c_loc(rec_u), c_loc(rec_v), c_loc(mnmx), thin, &
                                cmsg, int(lstr, c_int) )
  end if
  call gpu_check( rc, cmsg, 'beom_download_outputs' )
  h0_on_gpu = .true.
  call put_record_r4( 'eta_.bin', out_rec, rec_eta, .false. )
  call put_record_r4( 'u___.bin', out_rec, rec_u,   .false. )
  call put_record_r4( 'v___.bin', out_rec, rec_v,   .false. )
  if ( diag > 0.5_rw ) then                                      ! pvor, mont, v_cc (:2884-2974) formed on the device too
    if ( ngpu > 1 ) then
      rc = beom_multi_download_diag( gpus, c_loc(rec_eta), c_loc(rec_u), c_loc(rec_v), cmsg, int(lstr, c_int) )
    else
      rc = beom_download_diag( gpu, c_loc(rec_eta), c_loc(rec_u), c_loc(rec_v), cmsg, int(lstr, c_int) )
    end if
    call gpu_check( rc, cmsg, 'beom_download_diag' )
    call put_record_r4( 'pvor.bin', out_rec, rec_eta, .false. )
    call put_record_r4( 'mont.bin', out_rec, rec_u,   .false. )
    call put_record_r4( 'v_cc.bin', out_rec, rec_v,   .false. )
  end if
  unum = get_un()
  open( unit = unum, file = trim(odir) // 'time.txt', form = 'formatted', action = 'write', status = 'old', &
        position = 'append', iostat = ios )
  write( unum, * ) real(ctim, r8)
  close( unum )
  out_rec = out_rec + 1
  write(ioso, *) 'ctim = ', ctim, ' days; dt_s = ', dt_s, ' days; record = ', out_rec - 1
  do k = 1, nlay
    write(ioso, *) 'min/max h', k, '= ', real(mnmx(1, k), rw), real(mnmx(2, k), rw)
    write(ioso, *) 'min/max u', k, '= ', real(mnmx(3, k), rw), real(mnmx(4, k), rw)
    write(ioso, *) 'min/max v', k, '= ', real(mnmx(5, k), rw), real(mnmx(6, k), rw)
  end do
  if ( thin > 0 ) then
    write(txt, '(1i8)') thin
    call fail( -int(thin), 'layer number ' // trim(txt) // ' has its thickness < hmin; Calculation halted.' )
  end if
end subroutine write_outputs_from_gpu

subroutine write_field(var, fresh)                               ! write_array (:2817-3001)
  character(4), intent(in) :: var
  logical, intent(in)      :: fresh
  real(r4), allocatable    :: a(:,:)
  real(rw), allocatable    :: w1(:,:), w2(:,:)
  real(rw) :: zeta
  integer  :: k, n, m, c1, c2, c3, c5, c6, c7
  allocate( a(ndeg, nlay) )
  select case ( var )
  case ( 'eta_' )                                                ! interface elevation, bottom-up
    do k = nlay, 1, -1
      do n = 1, ndeg
        if ( k == nlay ) then
          a(n, k) = real( hlay(n, k) - real(h0r4(n, k), r8), r4 )
        else
          a(n, k) = real( hlay(n, k) - real(h0r4(n, k), r8) + real(a(n, k+1), r8), r4 )
        end if
      end do
    end do
    if ( rgld > 0.5_rw ) a(:, 1) = real( lid_p(1:) )             ! the lid pressure in the top record (:2864-2872)
  case ( 'u___' )
    a = real( u(1:, :), r4 )
  case ( 'v___' )
    a = real( v(1:, :), r4 )
  case ( 'v_cc' )                                                ! Leith viscosity diagnosed from u, v (:2884-2929)
    allocate( w1(0:ndeg, nlay), w2(0:ndeg, nlay) )
    a = 0._r4;  w1 = 0._rw;  w2 = 0._rw
    do k = 1, nlay
      do n = 1, ndeg
        c1 = neig(1,n); c3 = neig(3,n); c5 = neig(5,n); c7 = neig(7,n)
        w1(n, k) = ( ( v(n,k) - v(c5,k) ) / dl - ( u(n,k) - u(c7,k) ) / dl ) * mkpe(n)
        w2(n, k) = ( u(c1,k) - u(n,k) ) / dl + ( v(c3,k) - v(n,k) ) / dl
      end do
      do n = 1, ndeg
        c1 = neig(1,n); c2 = neig(2,n); c3 = neig(3,n); c5 = neig(5,n); c7 = neig(7,n)
        a(n, k) = real( bvis + dvis * dl**2 * sqrt( ( w1(c1,k) - w1(n,k) )**2 + ( w1(c2,k) - w1(c3,k) )**2 &
                      + ( w1(c3,k) - w1(n,k) )**2 + ( w1(c2,k) - w1(c1,k) )**2 + ( w2(c1,k) - w2(n,k) )**2 &
                      + ( w2(n,k) - w2(c5,k) )**2 + ( w2(c3,k) - w2(n,k) )**2 + ( w2(n,k) - w2(c7,k) )**2 ), r4 )
      end do
    end do
    deallocate( w1, w2 )
  case ( 'mont' )                                                ! (:2930-2950)
    a = 0._r4
    do k = 1, nlay
      do n = 1, ndeg
        a(n, k) = real( - ocrp / real(nsal - 1, rw) * hsal * mk_n(n) &
                        * ( hsal / ( hmin * (1._rw - mk_n(n)) + real(hlay(n,k), rw) ) )**(nsal - 1), r4 )
        do m = 1, k - 1
          a(n, k) = a(n, k) - real( ( rhon(k) - rhon(m) ) * real(hlay(n,m), rw) / rhon(k), r4 )
        end do
        a(n, k) = a(n, k) + real( sum( hlay(n, :) ) - real(h_th(n), r8), r4 )
      end do
    end do
  case ( 'pvor' )                                                ! (:2951-2974)
    a = 0._r4
    do k = 1, nlay
      do n = 1, ndeg
        c5 = neig(5,n); c6 = neig(6,n); c7 = neig(7,n)
        zeta = ( ( v(n,k) - v(c5,k) ) / dl - ( u(n,k) - u(c7,k) ) / dl ) * mkpe(n)
        a(n, k) = real( ( fcor(n) + zeta * uadv ) * mkpi(n) * ( mk_n(n) + mk_n(c5) + mk_n(c7) + mk_n(c6) ) &
                        / real( hlay(n,k) + hlay(c5,k) + hlay(c6,k) + hlay(c7,k), rw ), r4 )
      end do
    end do
  end select
  call put_record_r4( var // '.bin', out_rec, a, fresh )
  deallocate( a )
end subroutine write_field

! ---- GPU hand-over -------------------------------------------------------------------
subroutine gpu_start()
  type(beom_params) :: prm
  character(kind = c_char) :: cmsg(lstr + 1)
  type(c_ptr) :: p_hdot, p_tide, p_bodf
  integer(c_int) :: rc
  integer(c_int), allocatable :: devs(:)
  character(len = 16) :: envv
  integer :: ios, k
  prm%abi_version = beom_abi_ver
  prm%lm = lm;  prm%mm = mm;  prm%nlay = nlay;  prm%ndeg = ndeg;  prm%nsal = nsal
  prm%variant = 0
  prm%flag_nudging = merge(1, 0, nudging_on)
  prm%dense_hint = 1;  prm%slab_row0 = 0;  prm%slab_mm = 0
  prm%dl = dl;  prm%dt = dt;  prm%grav = grav;  prm%rho0 = rho0
  prm%beta = beta;  prm%epsi = epsi;  prm%gamm = gamm;  prm%del1 = del1;  prm%del2 = del2
  prm%hmin = hmin;  prm%hsal = hsal;  prm%bvis = bvis;  prm%dvis = dvis;  prm%svis = FORK_SVIS
  prm%bdrg = bdrg;  prm%tdrg = FORK_TDRG;  prm%qdrg = qdrg;  prm%hsbl = hsbl;  prm%hbbl = hbbl
  prm%g_fb = g_fb;  prm%uadv = uadv;  prm%ocrp = ocrp;  prm%rgld = rgld;  prm%mcbc = mcbc
  prm%invf = invf;  prm%w_ti = w_ti(1)
  prm%rhon = 0._rw
  prm%rhon(1:nlay) = rhon(1:nlay)
  p_hdot = c_null_ptr;  p_tide = c_null_ptr;  p_bodf = c_null_ptr
  if ( has_hdot ) p_hdot = c_loc(hdot)
  if ( has_tide ) p_tide = c_loc(tide)
  if ( has_bodf ) p_bodf = c_loc(bodf)
  cmsg = c_null_char
  call get_environment_variable( 'BEOM_NGPU', envv, status = ios )
  ngpu = 1
  if ( ios == 0 ) read( envv, *, iostat = ios ) ngpu
  if ( ios /= 0 .or. ngpu < 1 ) ngpu = 1
  if ( ngpu > 1 ) then                                            ! one process, ngpu devices: bands of rows
    allocate( devs(ngpu) )
    do k = 1, ngpu
      devs(k) = int(k - 1, c_int)
    end do
    rc = beom_multi_create( prm, int(ngpu, c_int), devs, c_loc(neig), c_loc(subc), c_loc(mk_u), c_loc(mk_v),  &
                            c_loc(mk_n), c_loc(mkpe), c_loc(mkpi), c_loc(fcor), c_loc(h_th), c_loc(h_to),      &
                            c_loc(nudg), c_loc(fnud), p_hdot, p_tide, p_bodf, c_loc(taus), gpus, cmsg,         &
                            int(lstr, c_int) )
    call gpu_check( rc, cmsg, 'beom_multi_create' )
    if ( nudging_on .and. mcbc < 0.5_rw ) then                   ! no_gradient_obc (:2613-2679): segments dealt to the bands
      rc = beom_multi_set_open_boundaries( gpus, int(size(segm, 1), c_int), c_loc(segm), cmsg, int(lstr, c_int) )
      call gpu_check( rc, cmsg, 'beom_multi_set_open_boundaries' )
    end if
    rc = beom_multi_upload_state( gpus, c_loc(hlay), c_loc(u), c_loc(v), c_loc(h_u), c_loc(h_v), c_loc(rs_h), &
                                  c_loc(dmdx), c_loc(dmdy), c_loc(v_cc), c_loc(v_ll), c_loc(tt3d),           &
                                  c_loc(tb3d), c_loc(tu3d), cmsg, int(lstr, c_int) )
    call gpu_check( rc, cmsg, 'beom_multi_upload_state' )
    write(ioso, *) 'MI355X engine: ', ngpu, ' devices, row bands with ghost exchange.'
    deallocate( devs )
    return
  end if
  rc = beom_create( prm, 0_c_int, c_loc(neig), c_loc(subc), c_loc(mk_u), c_loc(mk_v), c_loc(mk_n), &
                    c_loc(mkpe), c_loc(mkpi), c_loc(fcor), c_loc(h_th), c_loc(h_to), c_loc(nudg),  &
                    c_loc(fnud), p_hdot, p_tide, p_bodf, c_loc(taus), gpu, cmsg, int(lstr, c_int) )
  call gpu_check( rc, cmsg, 'beom_create' )
  if ( nudging_on .and. mcbc < 0.5_rw ) then                     ! no_gradient_obc (:2613-2679) on the device
    rc = beom_set_open_boundaries( gpu, int(size(segm, 1), c_int), c_loc(segm), cmsg, int(lstr, c_int) )
    call gpu_check( rc, cmsg, 'beom_set_open_boundaries' )
  end if
  if ( rgld > 0.5_rw ) then                                      ! surf_pressure (:1705-1838) on the device
    rc = beom_set_rigid_lid( gpu, c_loc(lid_w), c_loc(lid_s), c_loc(lid_inv), c_loc(lid_p), cmsg, int(lstr, c_int) )
    call gpu_check( rc, cmsg, 'beom_set_rigid_lid' )
  end if
  rc = beom_upload_state( gpu, c_loc(hlay), c_loc(u), c_loc(v), c_loc(h_u), c_loc(h_v), c_loc(rs_h), &
                          c_loc(dmdx), c_loc(dmdy), c_loc(v_cc), c_loc(v_ll), c_loc(tt3d),           &
                          c_loc(tb3d), c_loc(tu3d), cmsg, int(lstr, c_int) )
  call gpu_check( rc, cmsg, 'beom_upload_state' )
  if ( beom_is_dense(gpu) == 1 ) then
    write(ioso, *) 'MI355X engine: dense-frame fast path.'
  else
    write(ioso, *) 'MI355X engine: packed gather path.'
  end if
end subroutine gpu_start

! ---- time loop = integrate_time (:1840-1919); the steps run on the GPU ---------------
subroutine advance()
  character(kind = c_char) :: cmsg(lstr + 1)
  real(r8) :: dtd8
  integer  :: nstp, notp, n_3d, first, last, lerm
  integer(c_int) :: rc

  lerm = len_trim(errm)
  errm = trim(errm) // ' in subroutine advance of the MI355X host (beom_host_mod.f95),'
  write(ioso, *) 'dl = ', dl, ' meters.'
  write(ioso, *) 'dt = ', dt, ' seconds.'
  dtd8 = real(dt, r8) / 24._r8 / 3600._r8
  nstp = nint( real(dt_s, r8) / dtd8 )
  notp = max( nint( real(dt_o, r8) / dtd8 ), 1 )
  n_3d = max( nint( real(dt3d, r8) / dtd8 ), 1 )
  cmsg = c_null_char

  ! steps 1-3: plain forward-backward, never followed by an output (:1861-1875)
  if ( ngpu > 1 ) then
    rc = beom_multi_step( gpus, 1_c_int, 3_c_int, tres, dtd8, real(dt_r, c_double), real(rsta, c_double), &
                          int(n_3d, c_int), cmsg, int(lstr, c_int) )
  else
    rc = beom_step( gpu, 1_c_int, 3_c_int, tres, dtd8, real(dt_r, c_double), real(rsta, c_double), &
                    int(n_3d, c_int), cmsg, int(lstr, c_int) )
  end if
  call gpu_check( rc, cmsg, 'beom_step' )

  first = 4
  do while ( first <= nstp )
    last = min( ((first + notp - 1) / notp) * notp, nstp )       ! run up to the next output step
    if ( ngpu > 1 ) then
      rc = beom_multi_step( gpus, int(first, c_int), int(last - first + 1, c_int), tres, dtd8,     &
                            real(dt_r, c_double), real(rsta, c_double), int(n_3d, c_int), cmsg, int(lstr, c_int) )
    else
      rc = beom_step( gpu, int(first, c_int), int(last - first + 1, c_int), tres, dtd8,            &
                      real(dt_r, c_double), real(rsta, c_double), int(n_3d, c_int), cmsg, int(lstr, c_int) )
    end if
    call gpu_check( rc, cmsg, 'beom_step' )
    if ( mod(last, notp) == 0 ) then
      ctim = real( tres + dtd8 * real(last, r8), rw )
      call write_outputs_from_gpu()                              ! only real*4 records cross PCIe (one or several devices)
    end if
    first = last + 1
  end do
  if ( ngpu > 1 ) then
    rc = beom_multi_sync( gpus, cmsg, int(lstr, c_int) )
  else
    rc = beom_sync( gpu, cmsg, int(lstr, c_int) )
  end if
  call gpu_check( rc, cmsg, 'beom_sync' )
  if ( errc /= 0 ) call quit()
  errm = errm(1:lerm)
end subroutine advance

end module private_mod
