! ltrans_hip_driver.f90 -- Fortran host harness around the MI355X engine.
!
! Plays the part of LTRANS' PROGRAM main for the hot path only: it reads the
! run's namelist file in the LTRANS.data syntax, owns par(numpar,13) and the
! other arrays the reference's modules own, and runs the reference's loops
!     run_LTRANS / run_External_Timestep / run_Internal_Timestep
!     (Model/LTRANS.f90:141-161, 548-612)
! calling lt_step() where the reference calls update_particles() (:596) and
! lt_set_slab() where updateHydro scatters a new record (:559,
! hydrodynamic_module.f90:1052-1410).  NetCDF is not available in this image,
! so grid / boundary / particle / history arrays come from a flat binary "case"
! file written by ltrans_amd.casefile (the arrays are exactly those initGrid,
! createBounds, setEle_all and initHydro produce).
!
! usage: ltrans_hip_driver <LTRANS.data> <case.bin> <out.bin> [n_external] [max_internal] [pipeline]
!   pipeline = 1: updateHydro through lt_slab_prefetch / lt_slab_commit instead of lt_set_slab
! Like the reference it appends every particle error to ErrorLog.txt (LTRANS.f90:858-877 etc., formats
! :761-775) and, with WriteModelTiming = .TRUE., writes Timing.csv (:471-476, :1669-1689) -- both in the
! current directory.
!
! Several GPUs (SURVEY 8e): one process per GPU, started by whatever starts processes (a shell loop, srun, mpirun)
! with the environment
!     LT_NRANKS = number of processes, LT_RANK = 0 .. LT_NRANKS-1, LT_COMM_FILE = a path all of them can reach
!     (unique per run), LT_DEVICE = the GPU of this process (default: LT_RANK)
! Rank g keeps particles [g*N/G, (g+1)*N/G) of the case file (contiguous blocks, so that a shard's place in the
! serial random stream is one offset), the fields are replicated, and the only exchange is the all-gather of the
! new (x,y,z) that the reference needs as one array at print steps (LTRANS.f90:604-612): lt_comm_init_file /
! lt_allgather_positions, RCCL inside the library.  With turbulence the shards take static windows of the stream
! (lt_set_rng_window_static) while no particle can stop or start drawing; a run with settlement, mortality, an
! open boundary, behaviours 1-5 or particles released after the start exchanges the shards' draw counts before every
! step instead (lt_exchange_rng_window: one int64 per rank over the same communicator, inside the library; the skip
! rule is LTRANS.f90:790-816, the one serial stream random_module.f90:166-197).  Every rank writes
! <out.bin>.rank<g> (its shard, then the gathered xyz(N,3)); ErrorLog.txt / Timing.csv get the same suffix on
! ranks > 0.  A case file with habitat polygons (settlementon): createPolySpecs through lt_create_polyspecs.
program ltrans_hip_driver
  use iso_c_binding
  use ltrans_hip
  implicit none

  ! ---- namelist variables (names are the LTRANS.data keys; types per the
  !      reference's input schema: hc and days are REAL)
  integer :: numpar
  real :: days
  integer :: iprint, dt, idt
  integer :: us, ws, tdim, Vtransform
  real :: hc
  double precision :: z0, constZeta, constSalt, constTemp, constU, constV, constW, constAks, constDens
  logical :: readZeta, readSalt, readTemp, readU, readV, readW, readAks, readDens
  logical :: HTurbOn, VTurbOn
  logical :: late_release, dynamic_window
  integer(c_int32_t) :: hab_hdr(6)
  real(c_double), allocatable, target :: polys(:,:), holes(:,:), maxbdis(:), maxhdis(:)
  type(lt_habitat) :: hab
  integer :: npoly, nhole
  double precision :: ConstantHTurb
  integer :: Behavior
  logical :: OpenOceanBoundary, mortality
  double precision :: deadage, pediage, swimstart, swimslow, swimfast, Sgradient, sink, Hswimspeed, Swimdepth
  double precision :: twistart, twiend, daylength, Em, Kd, thresh
  logical :: settlementon, holesExist
  integer :: minpolyid, maxpolyid, minholeid, maxholeid, pedges, hedges
  double precision :: PI, Earth_Radius, lonmin, latmin
  logical :: SphericalProjection
  integer :: seed, ErrorFlag, ijbuff
  logical :: BoundaryBLNs, SaltTempOn, TrackCollisions, WriteHeaders, WriteModelTiming, FreeSlip

  namelist/numparticles/ numpar
  namelist/timeparam/ days, iprint, dt, idt
  namelist/hydroparam/ us, ws, tdim, hc, z0, Vtransform, readZeta, constZeta, readSalt, constSalt, &
                       readTemp, constTemp, readU, constU, readV, constV, readW, constW, readAks,   &
                       constAks, readDens, constDens
  namelist/turbparam/ HTurbOn, VTurbOn, ConstantHTurb
  namelist/behavparam/ Behavior, OpenOceanBoundary, mortality, deadage, pediage, swimstart, swimslow, &
                       swimfast, Sgradient, sink, Hswimspeed, Swimdepth
  namelist/dvmparam/ twistart, twiend, daylength, Em, Kd, thresh
  namelist/settleparam/ settlementon, holesExist, minpolyid, maxpolyid, minholeid, maxholeid, pedges, hedges
  namelist/convparam/ PI, Earth_Radius, SphericalProjection, lonmin, latmin
  namelist/other/ seed, ErrorFlag, BoundaryBLNs, SaltTempOn, TrackCollisions, WriteHeaders, &
                  WriteModelTiming, ijbuff, FreeSlip

  ! ---- case data
  integer(c_int32_t) :: magic, vi, uj, c_us, c_ws, rho_nodes, u_nodes, v_nodes, rho_elements, &
                        u_elements, v_elements, nbounds, maxbound, maxisland, numislands, c_numpar, nrec
  real(c_double), allocatable, target :: rx(:), ry(:), ux(:), uy(:), vx(:), vy(:), depth(:), rho_angle(:)
  integer(c_int32_t), allocatable, target :: RE(:,:), UE(:,:), VE(:,:), r_Adj(:,:), u_Adj(:,:), v_Adj(:,:)
  real(c_double), allocatable, target :: SC(:), CS(:), SCW(:), CSW(:)
  real(c_double), allocatable, target :: bnd_x(:,:), bnd_y(:,:), bx(:), by(:), hx(:), hy(:)
  integer(c_int32_t), allocatable, target :: land(:), hid(:)
  real(c_double), allocatable, target :: par(:,:)
  integer(c_int32_t), allocatable, target :: P_r_element(:), P_u_element(:), P_v_element(:)
  integer(c_int32_t), allocatable :: loc_r(:), loc_u(:), loc_v(:)
  integer(c_int32_t) :: loc_err, loc_par
  integer(c_int32_t), allocatable, target :: P_behave(:), dead(:), oob(:), settle(:), endpoly(:), &
                                            hitBottom(:), hitLand(:), err_code(:)
  real(c_double), allocatable, target :: P_swim(:,:), P_pediage(:), P_deadage(:), settletime(:)
  real(c_double), allocatable, target :: zeta(:), Uvel(:,:), Vvel(:,:), Wvel(:,:), Kh(:,:)

  type(c_ptr) :: eng
  type(lt_config) :: cfg
  type(lt_grid) :: g
  type(lt_bounds) :: b
  type(lt_particles) :: pp
  type(lt_slab) :: slab
  type(lt_step_stats) :: st
  double precision :: ex(3), ix(3)
  integer :: p, it, stepT, stepIT, t_b, t_c, t_f, stepf, n, nargs, n_external, max_internal, ios, pipeline
  type(lt_rom_record) :: rec
  type(lt_output) :: outp
  real(c_double), allocatable, target :: o_lon(:), o_lat(:), o_z(:), o_age(:), o_status(:)
  integer(c_int64_t) :: total_errors
  character(len=512) :: nmlfile, casefile, outfile, arg
  type(lt_particles) :: perr
  real(c_double) :: ms4(4)
  double precision :: t_hydro, t_elapsed, tc(7)
  integer(8) :: clk0, clk1, clk_rate, clk_print
  integer :: printdt
  ! ---- sharding (one process per GPU)
  integer :: nranks, myrank, mydev, n0, n1, numpar_global, g_r, envlen, envstat
  character(len=512) :: commfile, sfx, errlog, timingcsv
  integer(c_int32_t), allocatable :: counts(:)
  real(c_double), allocatable, target :: xyz_all(:,:)
  real(c_double), allocatable :: par_g(:,:)
  integer(c_int32_t), allocatable :: er_g(:), eu_g(:), ev_g(:)

  nargs = command_argument_count()
  if (nargs < 3) then
    write(*,*) 'usage: ltrans_hip_driver <LTRANS.data> <case.bin> <out.bin> [n_external] [max_internal] [pipeline]'
    stop 2
  endif
  call get_command_argument(1, nmlfile)
  call get_command_argument(2, casefile)
  call get_command_argument(3, outfile)
  n_external = -1
  max_internal = -1
  pipeline = 0
  if (nargs >= 6) then
    call get_command_argument(6, arg); read(arg, *) pipeline
  endif
  if (nargs >= 4) then
    call get_command_argument(4, arg); read(arg, *) n_external
  endif
  if (nargs >= 5) then
    call get_command_argument(5, arg); read(arg, *) max_internal
  endif
  nranks = 1; myrank = 0; mydev = -1; commfile = ' '; sfx = ' '
  call get_environment_variable('LT_NRANKS', arg, envlen, envstat)
  if (envstat == 0 .and. envlen > 0) read(arg, *) nranks
  call get_environment_variable('LT_RANK', arg, envlen, envstat)
  if (envstat == 0 .and. envlen > 0) read(arg, *) myrank
  call get_environment_variable('LT_DEVICE', arg, envlen, envstat)
  if (envstat == 0 .and. envlen > 0) then
    read(arg, *) mydev
  else if (nranks > 1) then
    mydev = myrank
  endif
  call get_environment_variable('LT_COMM_FILE', commfile, envlen, envstat)
  if (envstat /= 0 .or. envlen == 0) commfile = ' '
  if (nranks < 1 .or. myrank < 0 .or. myrank >= nranks) stop 'LT_RANK / LT_NRANKS out of range'
  if (len_trim(commfile) > 0) then
    write(sfx, '(a,i0)') '.rank', myrank
    outfile = trim(outfile)//trim(sfx)
  else if (nranks > 1) then
    error stop 'LT_NRANKS > 1 needs LT_COMM_FILE'
  endif
  errlog = 'ErrorLog.txt'; timingcsv = 'Timing.csv'
  if (myrank > 0) then
    errlog = trim(errlog)//trim(sfx); timingcsv = trim(timingcsv)//trim(sfx)
  endif

  ! ---- getParams: same groups, same order as the reference's reader
  !      (parameter_module.f90:29-86); groups this path does not use are skipped
  HTurbOn = .false.; VTurbOn = .false.; settlementon = .false.; holesExist = .false.
  OpenOceanBoundary = .false.; mortality = .false.; SaltTempOn = .false.; FreeSlip = .false.
  TrackCollisions = .false.; Behavior = 0; seed = 9; ErrorFlag = 1; PI = 3.14159265358979d0
  open(1, file=trim(nmlfile), status='old', action='read', iostat=ios)
  if (ios /= 0) then
    write(*,*) 'cannot open ', trim(nmlfile); stop 2
  endif
  read(1, nml=numparticles)
  read(1, nml=timeparam)
  read(1, nml=hydroparam)
  read(1, nml=turbparam)
  read(1, nml=behavparam)
  read(1, nml=dvmparam, iostat=ios)
  if (ios /= 0) rewind(1)
  read(1, nml=settleparam)
  read(1, nml=convparam)
  rewind(1)
  read(1, nml=other)
  close(1)

  ! ---- the arrays of initGrid / createBounds / setEle_all / initHydro
  open(2, file=trim(casefile), access='stream', form='unformatted', status='old', action='read')
  read(2) magic, vi, uj, c_us, c_ws, rho_nodes, u_nodes, v_nodes, rho_elements, u_elements, &
          v_elements, nbounds, maxbound, maxisland, numislands, c_numpar, nrec
  if (magic /= 1280590641 .and. magic /= 1280590642) stop 'not an LTC1 / LTC2 case file'
  if (c_us /= us .or. c_ws /= ws .or. c_numpar /= numpar) then
    write(*,*) 'case file does not match the namelists: us,ws,numpar =', c_us, c_ws, c_numpar
    stop 2
  endif
  allocate(rx(rho_nodes), ry(rho_nodes), ux(u_nodes), uy(u_nodes), vx(v_nodes), vy(v_nodes))
  allocate(depth(rho_nodes), rho_angle(rho_nodes))
  allocate(RE(4,rho_elements), UE(4,u_elements), VE(4,v_elements))
  allocate(r_Adj(rho_elements,10), u_Adj(u_elements,10), v_Adj(v_elements,10))
  allocate(SC(us), CS(us), SCW(ws), CSW(ws))
  allocate(bnd_x(2,nbounds), bnd_y(2,nbounds), land(nbounds), bx(maxbound), by(maxbound))
  allocate(hx(max(maxisland,1)), hy(max(maxisland,1)), hid(max(maxisland,1)))
  allocate(par(numpar,13), P_r_element(numpar), P_u_element(numpar), P_v_element(numpar))
  read(2) rx, ry, ux, uy, vx, vy, depth, rho_angle
  read(2) RE, UE, VE, r_Adj, u_Adj, v_Adj
  read(2) SC, CS, SCW, CSW
  read(2) bnd_x, bnd_y, land, bx, by
  if (maxisland > 0) read(2) hx(1:maxisland), hy(1:maxisland), hid(1:maxisland)
  read(2) par, P_r_element, P_u_element, P_v_element
  if (magic == 1280590642) then
    ! initSettlement's tables (settlement_module.f90:98-240): the polygons' and holes' edge points
    read(2) hab_hdr
    if (hab_hdr(1) /= pedges .or. hab_hdr(2) /= hedges .or. hab_hdr(3) /= minpolyid .or. hab_hdr(4) /= maxpolyid) &
      stop 'case file habitat does not match $settleparam'
    npoly = maxpolyid - minpolyid + 1
    nhole = 0
    if (hedges > 0) nhole = maxholeid - minholeid + 1
    allocate(polys(pedges,5), holes(max(hedges,1),6), maxbdis(npoly), maxhdis(max(nhole,1)))
    read(2) polys
    if (hedges > 0) read(2) holes
    read(2) maxbdis
    if (nhole > 0) read(2) maxhdis(1:nhole)
  else if (settlementon) then
    stop 'settlementon needs a case file with habitat polygons (LTC2)'
  endif
  allocate(zeta(rho_nodes), Uvel(u_nodes,us), Vvel(v_nodes,us), Wvel(rho_nodes,ws), Kh(rho_nodes,ws))
  ! (a particle of ANY rank that is released later -- LTRANS.f90:790: ix(3) <= par(n,pDOB) -> cycle, before any
  ! draw -- does not draw until then: the draw counts of the job change during the run)
  late_release = any(par(:,11) > 0.0_c_double)
  ! ---- this rank's contiguous block of the particles (everything below sees numpar = the shard)
  numpar_global = numpar
  allocate(counts(nranks))
  do g_r = 0, nranks - 1
    counts(g_r + 1) = int((int(g_r + 1, 8) * numpar_global) / nranks - (int(g_r, 8) * numpar_global) / nranks, c_int32_t)
  enddo
  n0 = int((int(myrank, 8) * numpar_global) / nranks)
  n1 = int((int(myrank + 1, 8) * numpar_global) / nranks)
  if (nranks > 1) then
    call move_alloc(par, par_g); call move_alloc(P_r_element, er_g)
    call move_alloc(P_u_element, eu_g); call move_alloc(P_v_element, ev_g)
    numpar = n1 - n0
    allocate(par(numpar,13), P_r_element(numpar), P_u_element(numpar), P_v_element(numpar))
    par = par_g(n0+1:n1, :)
    P_r_element = er_g(n0+1:n1); P_u_element = eu_g(n0+1:n1); P_v_element = ev_g(n0+1:n1)
    deallocate(par_g, er_g, eu_g, ev_g)
  endif
  ! A sharded run with turbulence sets each shard's window of the serial random stream ONCE (below:
  ! lt_set_rng_window_static) while no particle of the job can start or stop drawing; otherwise the shards' draw counts
  ! are exchanged before every step (lt_exchange_rng_window).  Decided here from the case file, which every rank reads.
  dynamic_window = len_trim(commfile) > 0 .and. (HTurbOn .or. VTurbOn .or. (Behavior >= 1 .and. Behavior <= 5)) .and. &
                   (settlementon .or. mortality .or. OpenOceanBoundary .or. (Behavior >= 1 .and. Behavior <= 5) .or. late_release)

  ! ---- initBehave (behavior_module.f90:94-160) / initSettlement state
  allocate(P_behave(numpar), dead(numpar), oob(numpar), settle(numpar), endpoly(numpar))
  allocate(hitBottom(numpar), hitLand(numpar), err_code(numpar))
  allocate(P_swim(numpar,3), P_pediage(numpar), P_deadage(numpar), settletime(numpar))
  do n = 1, numpar
    P_behave(n) = Behavior
    P_pediage(n) = pediage
    P_deadage(n) = deadage
    P_swim(n,1) = (swimfast - swimslow)/(P_pediage(n) - swimstart)
    P_swim(n,2) = swimfast - P_swim(n,1)*P_pediage(n)
    P_swim(n,3) = 0.0
    settletime(n) = P_pediage(n)
  enddo
  dead = 0; oob = 0; settle = 0; endpoly = 0; hitBottom = 0; hitLand = 0; err_code = 0

  ! ---- hand everything to the device
  cfg = lt_config_init()
  cfg%numpar = numpar; cfg%dt = dt; cfg%idt = idt; cfg%us = us; cfg%ws = ws
  cfg%hc = hc; cfg%Vtransform = Vtransform; cfg%z0 = z0
  cfg%HTurbOn = merge(1, 0, HTurbOn); cfg%VTurbOn = merge(1, 0, VTurbOn)
  cfg%ConstantHTurb = ConstantHTurb
  cfg%Behavior = Behavior
  cfg%OpenOceanBoundary = merge(1, 0, OpenOceanBoundary); cfg%mortality = merge(1, 0, mortality)
  cfg%deadage = deadage; cfg%pediage = pediage; cfg%swimstart = swimstart
  cfg%swimslow = swimslow; cfg%swimfast = swimfast; cfg%sink = sink; cfg%Swimdepth = Swimdepth
  cfg%Sgradient = Sgradient; cfg%Hswimspeed = Hswimspeed
  cfg%twistart = twistart; cfg%twiend = twiend; cfg%daylength = daylength
  cfg%Em = Em; cfg%Kd = Kd; cfg%thresh = thresh
  cfg%settlementon = merge(1, 0, settlementon); cfg%holesExist = merge(1, 0, holesExist)
  cfg%PI = PI; cfg%seed = seed; cfg%ErrorFlag = ErrorFlag
  cfg%Earth_Radius = Earth_Radius; cfg%SphericalProjection = merge(1, 0, SphericalProjection)
  cfg%lonmin = lonmin; cfg%latmin = latmin
  cfg%SaltTempOn = merge(1, 0, SaltTempOn); cfg%TrackCollisions = merge(1, 0, TrackCollisions)
  cfg%FreeSlip = merge(1, 0, FreeSlip)
  cfg%device = mydev
  eng = c_null_ptr
  call lt_check(lt_create(cfg, eng), eng, 'lt_create')
  if (len_trim(commfile) > 0) then
    call lt_check(lt_comm_init_file(eng, trim(commfile)//c_null_char, int(nranks, c_int32_t), &
                                    int(myrank, c_int32_t), 120_c_int32_t), eng, 'lt_comm_init_file')
  endif
  if (nranks > 1 .and. (HTurbOn .or. VTurbOn) .and. .not. dynamic_window) then
    ! the shard's window of the serial stream: the tracked particles of the lower ranks come first
    call lt_check(lt_set_rng_window_static(eng, int(n0, c_int64_t), int(numpar_global, c_int64_t)), eng, &
                  'lt_set_rng_window_static')
  endif

  g%rho_nodes = rho_nodes; g%u_nodes = u_nodes; g%v_nodes = v_nodes
  g%rho_elements = rho_elements; g%u_elements = u_elements; g%v_elements = v_elements
  g%rx = c_loc(rx); g%ry = c_loc(ry); g%ux = c_loc(ux); g%uy = c_loc(uy)
  g%vx = c_loc(vx); g%vy = c_loc(vy); g%depth = c_loc(depth); g%rho_angle = c_loc(rho_angle)
  g%rho_mask = c_null_ptr; g%u_mask = c_null_ptr; g%v_mask = c_null_ptr
  g%RE = c_loc(RE); g%UE = c_loc(UE); g%VE = c_loc(VE)
  g%r_Adjacent = c_loc(r_Adj); g%u_Adjacent = c_loc(u_Adj); g%v_Adjacent = c_loc(v_Adj)
  g%SC = c_loc(SC); g%CS = c_loc(CS); g%SCW = c_loc(SCW); g%CSW = c_loc(CSW)
  call lt_check(lt_set_grid(eng, g), eng, 'lt_set_grid')

  b%nbounds = nbounds; b%maxbound = maxbound; b%maxisland = maxisland; b%numislands = numislands
  b%bnd_x = c_loc(bnd_x); b%bnd_y = c_loc(bnd_y); b%land = c_loc(land)
  b%bx = c_loc(bx); b%by = c_loc(by); b%hx = c_loc(hx); b%hy = c_loc(hy); b%hid = c_loc(hid)
  call lt_check(lt_set_bounds(eng, b), eng, 'lt_set_bounds')

  if (settlementon) then
    ! initSettlement -> createPolySpecs (settlement_module.f90:245-480) on the device
    hab%pedges = pedges; hab%hedges = hedges; hab%minpolyid = minpolyid; hab%maxpolyid = maxpolyid
    hab%minholeid = minholeid; hab%maxholeid = maxholeid
    hab%polys = c_loc(polys); hab%holes = c_loc(holes); hab%maxbdis = c_loc(maxbdis); hab%maxhdis = c_loc(maxhdis)
    hab%polyspecs = c_null_ptr; hab%holespecs = c_null_ptr; hab%elepolys_ptr = c_null_ptr; hab%elepolys_ids = c_null_ptr
    hab%polyholes_ptr = c_null_ptr; hab%polyholes_ids = c_null_ptr
    call lt_check(lt_create_polyspecs(eng, hab), eng, 'lt_create_polyspecs')
  endif

  ! setEle_all (LTRANS.f90:410, hydrodynamic_module.f90:1536-1676) on the device; the case file's
  ! elements (located by the test's host code) are kept only to report the comparison
  allocate(loc_r(numpar), loc_u(numpar), loc_v(numpar))
  call lt_check(lt_locate_particles(eng, int(numpar, c_int32_t), par(:,1), par(:,2), loc_r, loc_u, loc_v, &
                                    loc_err, loc_par), eng, 'lt_locate_particles')
  if (loc_err /= 0) then
    write(*,'(a,i0,a,i0)') ' setEle_all: error ', loc_err, ' at particle ', loc_par
    stop 3
  end if
  write(*,'(a,i0,a)') ' setEle_all on the device: ', &
    count(loc_r /= P_r_element .or. loc_u /= P_u_element .or. loc_v /= P_v_element), &
    ' particles located differently from the case file'
  P_r_element = loc_r; P_u_element = loc_u; P_v_element = loc_v

  pp%par = c_loc(par)
  pp%P_r_element = c_loc(P_r_element); pp%P_u_element = c_loc(P_u_element); pp%P_v_element = c_loc(P_v_element)
  pp%P_behave = c_loc(P_behave); pp%dead = c_loc(dead); pp%oob = c_loc(oob)
  pp%P_swim = c_loc(P_swim); pp%P_pediage = c_loc(P_pediage); pp%P_deadage = c_loc(P_deadage)
  pp%settle = c_loc(settle); pp%settletime = c_loc(settletime)
  pp%endpoly = c_loc(endpoly); pp%hitBottom = c_loc(hitBottom); pp%hitLand = c_loc(hitLand)
  pp%err_code = c_loc(err_code)
  ! the case file carries behaviours 0 and 6 only: no salinity-cue state
  pp%timer = c_null_ptr; pp%P_Sprev = c_null_ptr; pp%P_zprev = c_null_ptr; pp%bottom = c_null_ptr
  pp%P_Salt = c_null_ptr; pp%P_Temp = c_null_ptr
  call lt_check(lt_set_particles(eng, pp), eng, 'lt_set_particles')

  ! ---- initHydro: first three records into slots 1,2,3 (hydrodynamic_module.f90:741-743)
  slab%zeta = c_loc(zeta); slab%Uvel = c_loc(Uvel); slab%Vvel = c_loc(Vvel)
  slab%Wvel = c_loc(Wvel); slab%Kh = c_loc(Kh)
  slab%salt = c_null_ptr; slab%temp = c_null_ptr
  do n = 1, 3
    read(2) zeta, Uvel, Vvel, Wvel, Kh
    call lt_check(lt_set_slab(eng, n, slab), eng, 'lt_set_slab')
  enddo
  t_b = 1; t_c = 2; t_f = 3; stepf = 3
  call lt_check(lt_set_time_slots(eng, t_b, t_c, t_f), eng, 'lt_set_time_slots')

  ! ErrorLog.txt is opened fresh by ini_LTRANS (LTRANS.f90:337-354); only err_code comes back for it
  open(210, file=trim(errlog), status='replace'); close(210)
  perr%par = c_null_ptr
  perr%P_r_element = c_null_ptr; perr%P_u_element = c_null_ptr; perr%P_v_element = c_null_ptr
  perr%P_behave = c_null_ptr; perr%dead = c_null_ptr; perr%oob = c_null_ptr
  perr%P_swim = c_null_ptr; perr%P_pediage = c_null_ptr; perr%P_deadage = c_null_ptr
  perr%settle = c_null_ptr; perr%settletime = c_null_ptr
  perr%endpoly = c_null_ptr; perr%hitBottom = c_null_ptr; perr%hitLand = c_null_ptr
  perr%timer = c_null_ptr; perr%P_Sprev = c_null_ptr; perr%P_zprev = c_null_ptr; perr%bottom = c_null_ptr
  perr%P_Salt = c_null_ptr; perr%P_Temp = c_null_ptr
  perr%err_code = c_loc(err_code)
  if (WriteModelTiming) then        ! LTRANS.f90:471-476
    open(300, file=trim(timingcsv), status='replace')
    write(300,'(A)') 'Daytime,Elapsed,Hydro,Hydro %,Set Up,Set Up %,'// &
                     'Advection,Advection %,HTurb,HTurb %,VTurb,VTurb %,Behavior,'// &
                     'Behavior %,Update,Update %'
    close(300)
    call lt_check(lt_phase_timing(eng, 1_c_int32_t), eng, 'lt_phase_timing')
  endif
  call system_clock(clk_print, clk_rate)
  t_hydro = 0.0d0
  printdt = 0

  ! ---- run_LTRANS (LTRANS.f90:141-161)
  stepT = int(days*86400.0)/dt
  if (n_external > 0) stepT = min(stepT, n_external)
  stepIT = int(dt/idt)
  if (max_internal > 0) stepIT = min(stepIT, max_internal)
  total_errors = 0
  do p = 1, stepT
    ! run_External_Timestep (:548-579)
    if (p > 2) then
      ! updateHydro: rotate indices (hydrodynamic_module.f90:1080-1082), one new record into t_f
      t_b = mod(t_b,3)+1
      t_c = mod(t_c,3)+1
      t_f = mod(t_f,3)+1
      stepf = stepf + 1
      if (stepf > nrec) stop 'case file holds too few history records'
      read(2) zeta, Uvel, Vvel, Wvel, Kh
      call system_clock(clk0)
      if (pipeline == 0) then
        call lt_check(lt_set_slab(eng, t_f, slab), eng, 'lt_set_slab')
        call lt_check(lt_set_time_slots(eng, t_b, t_c, t_f), eng, 'lt_set_time_slots')
      else
        ! the same record handed over as updateHydro holds it (rom*f arrays are node-numbered already;
        ! the case file's values carry the mask multiply), copied and scattered asynchronously
        rec%struct_size = int(c_sizeof(rec), c_int32_t); rec%dtype = LT_ROM_F64
        rec%vi = vi; rec%uj = uj; rec%ui = vi - 1; rec%vj = uj - 1
        rec%masked = 1; rec%has_window = 0; rec%window = 0
        rec%romZ = c_loc(zeta); rec%romU = c_loc(Uvel); rec%romV = c_loc(Vvel); rec%romW = c_loc(Wvel)
        rec%romKH = c_loc(Kh); rec%romS = c_null_ptr; rec%romT = c_null_ptr
        call lt_check(lt_slab_prefetch(eng, rec), eng, 'lt_slab_prefetch')
        call lt_check(lt_slab_commit(eng, t_b, t_c, t_f), eng, 'lt_slab_commit')
      endif
      call system_clock(clk1)
      t_hydro = t_hydro + dble(clk1 - clk0) / dble(clk_rate)
    endif
    ex = 0.0
    ex(1) = (p-2)*dt
    ex(2) = (p-1)*dt
    ex(3) = p*dt
    do it = 1, stepIT
      ! run_Internal_Timestep (:583-612)
      ix(1) = ex(2) + DBLE((it-2)*idt)
      ix(2) = ex(2) + DBLE((it-1)*idt)
      ix(3) = ex(2) + DBLE(it*idt)
      ! the shards' draw counts of this step, exchanged inside the library (a particle that settled, died, left or
      ! has not been released draws nothing: LTRANS.f90:790-816), then the step itself
      if (dynamic_window) call lt_check(lt_exchange_rng_window(eng, ex, ix, p, it), eng, 'lt_exchange_rng_window')
      call lt_check(lt_step(eng, ex, ix, p, it), eng, 'lt_step')   ! = call update_particles()
      call lt_check(lt_get_stats(eng, st), eng, 'lt_get_stats')
      total_errors = total_errors + st%errors
      if (st%errors > 0) then
        ! the reference appends one line per event to ErrorLog.txt as it happens; here the step's
        ! per-particle codes come back once and are written in particle order (formats LTRANS.f90:761-775)
        call lt_check(lt_get_particles(eng, perr), eng, 'lt_get_particles(err_code)')
        open(210, file=trim(errlog), position='append')
        do n = 1, numpar
          select case (err_code(n))
            case (4);  write(210,21) n, int(ix(3))
            case (5);  write(210,22) n, int(ix(3))
            case (6);  write(210,23) n, int(ix(3))
            case (24); write(210,24) n, int(ix(3))
            case (25); write(210,25) n, int(ix(3))
            case (26); write(210,26) n, int(ix(3))
            case (27); write(210,27) n, int(ix(3))
            case (28); write(210,28) n, int(ix(3))
            case (29); write(210,29) n, int(ix(3))
          end select
        enddo
        close(210)
      endif
      if (st%errors > 0 .and. (ErrorFlag < 1 .or. ErrorFlag > 3)) then
        write(*,*) st%errors, ' particle error(s) at model time ', ix(3)
        write(*,*) 'The Program Cannot Continue and Will Terminate'
        stop 1
      endif
      ! printOutput's timing line (LTRANS.f90:1669-1689) every iprint seconds of model time
      printdt = printdt + idt
      if (printdt >= iprint) then
        printdt = 0
        if (WriteModelTiming) then
          call lt_check(lt_get_phase_times(eng, ms4, 1_c_int32_t), eng, 'lt_get_phase_times')
          call system_clock(clk1)
          t_elapsed = dble(clk1 - clk_print) / dble(clk_rate)
          clk_print = clk1
          tc = 0.0d0
          tc(1) = t_hydro; tc(2) = ms4(1) * 1.0d-3; tc(3) = ms4(2) * 1.0d-3
          tc(5) = ms4(3) * 1.0d-3; tc(7) = ms4(4) * 1.0d-3      ! HTurb, Behavior: inside the update kernel
          t_hydro = 0.0d0
          open(300, file=trim(timingcsv), position='append')
          write(300,"(15(F14.4,','),F14.4)") ix(3)/86400.0d0, t_elapsed, &
            tc(1), tc(1)/t_elapsed*100.0d0, tc(2), tc(2)/t_elapsed*100.0d0, tc(3), tc(3)/t_elapsed*100.0d0, &
            tc(4), tc(4)/t_elapsed*100.0d0, tc(5), tc(5)/t_elapsed*100.0d0, tc(6), tc(6)/t_elapsed*100.0d0, &
            tc(7), tc(7)/t_elapsed*100.0d0
          close(300)
        endif
      endif
    enddo
  enddo
  close(2)

  ! ---- fin: bring the state back and write it out
  call lt_check(lt_get_particles(eng, pp), eng, 'lt_get_particles')
  open(3, file=trim(outfile), access='stream', form='unformatted', status='replace', action='write')
  write(3) int(numpar, c_int32_t), int(stepT*stepIT, c_int32_t)
  write(3) par, P_r_element, P_u_element, P_v_element, err_code, hitBottom, hitLand, dead, oob, settle
  ! printOutput (LTRANS.f90:1617): lon / lat / depth / age / status converted on the device
  allocate(o_lon(numpar), o_lat(numpar), o_z(numpar), o_age(numpar), o_status(numpar))
  outp%lon = c_loc(o_lon); outp%lat = c_loc(o_lat); outp%z = c_loc(o_z); outp%age = c_loc(o_age)
  outp%status = c_loc(o_status); outp%salt = c_null_ptr; outp%temp = c_null_ptr
  outp%hitBottom = c_null_ptr; outp%hitLand = c_null_ptr
  call lt_check(lt_get_output(eng, outp, 0_c_int32_t), eng, 'lt_get_output')
  write(3) o_lon, o_lat, o_z, o_age, o_status
  if (len_trim(commfile) > 0) then
    ! the print step's one array of all particles (LTRANS.f90:604-612): every rank receives xyz(N,3)
    allocate(xyz_all(numpar_global,3))
    call lt_check(lt_allgather_positions(eng, counts, xyz_all, c_null_ptr), eng, 'lt_allgather_positions')
    write(3) int(numpar_global, c_int32_t)
    write(3) xyz_all
    call lt_check(lt_comm_destroy(eng), eng, 'lt_comm_destroy')
  endif
  close(3)
  write(*,'(A,I0,A,I0,A,I0)') 'ltrans_hip_driver: ', numpar, ' particles, ', stepT*stepIT, &
        ' internal steps, particle errors: ', total_errors
  call lt_check(lt_destroy(eng), c_null_ptr, 'lt_destroy')
    21 FORMAT ('Particle ',I10,' not in rho element after ',I10,' seconds')
    22 FORMAT ('Particle ',I10,' not in u element after '  ,I10,' seconds')
    23 FORMAT ('Particle ',I10,' not in v element after '  ,I10,' seconds')
    24 FORMAT ('Particle ',I10,' out after 3rd reflection after ',I10,' seconds')
    25 FORMAT ('Particle ',I10,' outside main bounds after intersect_reflect after ',I10,' seconds')
    26 FORMAT ('Particle ',I10,' inside island bounds after intersect_reflect after ',I10,' seconds')
    27 FORMAT ('Particle ',I10,' jumped over rho element after ',I10,' seconds')
    28 FORMAT ('Particle ',I10,' jumped over u element after ',I10,' seconds')
    29 FORMAT ('Particle ',I10,' jumped over v element after ',I10,' seconds')
end program ltrans_hip_driver
