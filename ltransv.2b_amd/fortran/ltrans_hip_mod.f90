! ltrans_hip_mod.f90 -- ISO_C_BINDING interface to the MI355X engine
! (include/ltrans_hip.h, ABI version 1).
!
! This is the Fortran side of the drop-in boundary for LTRANS v.2b's
! update_particles() (Model/LTRANS.f90:707-1419): the host program stays
! Fortran; it hands the arrays the reference already owns to the device once
! (after initGrid / createBounds / initBehave / setEle_all / initHydro) and then
! calls lt_step(...) where the reference calls update_particles() (:596).
! Every TYPE below mirrors the C struct of the same name field for field;
! arrays are passed with C_LOC of the reference's own (column-major) arrays.
module ltrans_hip
  use iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: LT_ABI_EXPECTED = 2
  integer(c_int), parameter :: LT_OK = 0
  integer(c_int), parameter :: LT_SLAB_F32 = 0, LT_SLAB_F64 = 1

  type, bind(C) :: lt_config
    integer(c_int32_t) :: struct_size, abi_version
    integer(c_int32_t) :: numpar
    integer(c_int32_t) :: dt, idt
    integer(c_int32_t) :: us, ws
    real(c_float)      :: hc
    integer(c_int32_t) :: Vtransform
    real(c_double)     :: z0
    integer(c_int32_t) :: HTurbOn, VTurbOn
    real(c_double)     :: ConstantHTurb
    integer(c_int32_t) :: Behavior, OpenOceanBoundary, mortality
    real(c_double)     :: deadage, pediage, swimstart, swimslow, swimfast, sink, Swimdepth
    real(c_double)     :: Sgradient, Hswimspeed
    real(c_double)     :: twistart, twiend, daylength, Em, Kd, thresh
    integer(c_int32_t) :: settlementon, holesExist
    real(c_double)     :: PI, Earth_Radius
    integer(c_int32_t) :: SphericalProjection
    real(c_double)     :: lonmin, latmin
    integer(c_int32_t) :: seed, ErrorFlag, SaltTempOn, TrackCollisions, FreeSlip
    integer(c_int32_t) :: slab_storage, device, resort_every
  end type lt_config

  type, bind(C) :: lt_grid
    integer(c_int32_t) :: rho_nodes, u_nodes, v_nodes
    integer(c_int32_t) :: rho_elements, u_elements, v_elements
    type(c_ptr) :: rx, ry, ux, uy, vx, vy, depth, rho_angle
    type(c_ptr) :: rho_mask, u_mask, v_mask
    type(c_ptr) :: RE, UE, VE
    type(c_ptr) :: r_Adjacent, u_Adjacent, v_Adjacent
    type(c_ptr) :: SC, CS, SCW, CSW
  end type lt_grid

  type, bind(C) :: lt_bounds
    integer(c_int32_t) :: nbounds, maxbound, maxisland, numislands
    type(c_ptr) :: bnd_x, bnd_y, land, bx, by, hx, hy, hid
  end type lt_bounds

  type, bind(C) :: lt_habitat
    integer(c_int32_t) :: pedges, hedges, minpolyid, maxpolyid, minholeid, maxholeid
    type(c_ptr) :: polys, holes, polyspecs, holespecs, maxbdis, maxhdis
    type(c_ptr) :: elepolys_ptr, elepolys_ids, polyholes_ptr, polyholes_ids
  end type lt_habitat

  type, bind(C) :: lt_particles
    type(c_ptr) :: par
    type(c_ptr) :: P_r_element, P_u_element, P_v_element
    type(c_ptr) :: P_behave, dead, oob
    type(c_ptr) :: P_swim, P_pediage, P_deadage
    type(c_ptr) :: settle, settletime
    type(c_ptr) :: endpoly, hitBottom, hitLand, err_code
    type(c_ptr) :: timer, P_Sprev, P_zprev, bottom
    type(c_ptr) :: P_Salt, P_Temp
  end type lt_particles

  ! what writeOutput hands to the CSV / NetCDF writers (LTRANS.f90:1700-1779)
  type, bind(C) :: lt_output
    type(c_ptr) :: lon, lat, z, age, status
    type(c_ptr) :: salt, temp
    type(c_ptr) :: hitBottom, hitLand
  end type lt_output

  type, bind(C) :: lt_slab
    type(c_ptr) :: zeta, Uvel, Vvel, Wvel, Kh, salt, temp
  end type lt_slab

  ! a ROMS record as updateHydro holds it after NF90_GET_VAR, before the mask multiply
  ! (hydrodynamic_module.f90:1065-1368): rom*f arrays + t_ijruv
  integer(c_int32_t), parameter :: LT_ROM_F64 = 0, LT_ROM_F32 = 1
  type, bind(C) :: lt_rom_record
    integer(c_int32_t) :: struct_size, dtype
    integer(c_int32_t) :: vi, uj, ui, vj
    integer(c_int32_t) :: masked, has_window
    integer(c_int32_t) :: window(12)
    type(c_ptr) :: romZ, romU, romV, romW, romKH, romS, romT
  end type lt_rom_record

  type, bind(C) :: lt_step_stats
    integer(c_int64_t) :: active, errors, rng_draws
  end type lt_step_stats

  interface
    function lt_abi_version() bind(C, name="lt_abi_version") result(v)
      import :: c_int
      integer(c_int) :: v
    end function
    function lt_create(cfg, engine) bind(C, name="lt_create") result(rc)
      import :: c_int, c_ptr, lt_config
      type(lt_config), intent(in) :: cfg
      type(c_ptr), intent(out) :: engine
      integer(c_int) :: rc
    end function
    function lt_destroy(engine) bind(C, name="lt_destroy") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: engine
      integer(c_int) :: rc
    end function
    function lt_last_error(engine) bind(C, name="lt_last_error") result(msg)
      import :: c_ptr
      type(c_ptr), value :: engine
      type(c_ptr) :: msg
    end function
    function lt_set_grid(engine, g) bind(C, name="lt_set_grid") result(rc)
      import :: c_int, c_ptr, lt_grid
      type(c_ptr), value :: engine
      type(lt_grid), intent(in) :: g
      integer(c_int) :: rc
    end function
    function lt_set_bounds(engine, b) bind(C, name="lt_set_bounds") result(rc)
      import :: c_int, c_ptr, lt_bounds
      type(c_ptr), value :: engine
      type(lt_bounds), intent(in) :: b
      integer(c_int) :: rc
    end function
    function lt_set_habitat(engine, h) bind(C, name="lt_set_habitat") result(rc)
      import :: c_int, c_ptr, lt_habitat
      type(c_ptr), value :: engine
      type(lt_habitat), intent(in) :: h
      integer(c_int) :: rc
    end function
    function lt_set_particles(engine, p) bind(C, name="lt_set_particles") result(rc)
      import :: c_int, c_ptr, lt_particles
      type(c_ptr), value :: engine
      type(lt_particles), intent(in) :: p
      integer(c_int) :: rc
    end function
    function lt_get_particles(engine, p) bind(C, name="lt_get_particles") result(rc)
      import :: c_int, c_ptr, lt_particles
      type(c_ptr), value :: engine
      type(lt_particles), intent(in) :: p     ! pointers to the caller's arrays: filled in place
      integer(c_int) :: rc
    end function
    function lt_get_output(engine, o, reset_hits) bind(C, name="lt_get_output") result(rc)
      import :: c_int, c_int32_t, c_ptr, lt_output
      type(c_ptr), value :: engine
      type(lt_output), intent(in) :: o
      integer(c_int32_t), value :: reset_hits
      integer(c_int) :: rc
    end function
    function lt_set_slab(engine, slot, s) bind(C, name="lt_set_slab") result(rc)
      import :: c_int, c_int32_t, c_ptr, lt_slab
      type(c_ptr), value :: engine
      integer(c_int32_t), value :: slot
      type(lt_slab), intent(in) :: s
      integer(c_int) :: rc
    end function
    function lt_slab_prefetch(engine, r) bind(C, name="lt_slab_prefetch") result(rc)
      import :: c_int, c_ptr, lt_rom_record
      type(c_ptr), value :: engine
      type(lt_rom_record), intent(in) :: r
      integer(c_int) :: rc
    end function
    function lt_slab_commit(engine, t_b, t_c, t_f) bind(C, name="lt_slab_commit") result(rc)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: engine
      integer(c_int32_t), value :: t_b, t_c, t_f
      integer(c_int) :: rc
    end function
    function lt_set_time_slots(engine, t_b, t_c, t_f) bind(C, name="lt_set_time_slots") result(rc)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: engine
      integer(c_int32_t), value :: t_b, t_c, t_f
      integer(c_int) :: rc
    end function
    function lt_set_rng(engine, mt, mti) bind(C, name="lt_set_rng") result(rc)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: engine
      integer(c_int32_t), intent(in) :: mt(0:623)
      integer(c_int32_t), value :: mti
      integer(c_int) :: rc
    end function
    function lt_get_rng(engine, mt, mti) bind(C, name="lt_get_rng") result(rc)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: engine
      integer(c_int32_t), intent(out) :: mt(0:623)
      integer(c_int32_t), intent(out) :: mti
      integer(c_int) :: rc
    end function
    function lt_count_active(engine, ix, n_active) bind(C, name="lt_count_active") result(rc)
      import :: c_int, c_int64_t, c_double, c_ptr
      type(c_ptr), value :: engine
      real(c_double), intent(in) :: ix(3)
      integer(c_int64_t), intent(out) :: n_active
      integer(c_int) :: rc
    end function
    function lt_count_draws(engine, ex, ix, p, it, n_draws) bind(C, name="lt_count_draws") result(rc)
      import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
      type(c_ptr), value :: engine
      real(c_double), intent(in) :: ex(3), ix(3)
      integer(c_int32_t), value :: p, it
      integer(c_int64_t), intent(out) :: n_draws
      integer(c_int) :: rc
    end function
    function lt_set_rng_window_draws(engine, draws_before, draws_total) bind(C, name="lt_set_rng_window_draws") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: engine
      integer(c_int64_t), value :: draws_before, draws_total
      integer(c_int) :: rc
    end function
    ! the per-step exchange of the shards' draw counts (one int64 per rank over the communicator of lt_comm_init_file)
    ! and this shard's window of the serial random stream; the lt_step that follows takes the same ex, ix, p, it
    function lt_exchange_rng_window(engine, ex, ix, p, it) bind(C, name="lt_exchange_rng_window") result(rc)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), value :: engine
      real(c_double), intent(in) :: ex(3), ix(3)
      integer(c_int32_t), value :: p, it
      integer(c_int) :: rc
    end function
    function lt_set_rng_window(engine, active_before, active_total) &
        bind(C, name="lt_set_rng_window") result(rc)
      import :: c_int, c_int64_t, c_ptr
      type(c_ptr), value :: engine
      integer(c_int64_t), value :: active_before, active_total
      integer(c_int) :: rc
    end function
    ! = update_particles() with main's ex, ix, p, it (LTRANS.f90:568-571,588-590)
    function lt_set_rng_window_static(engine, active_before, active_total) &
        bind(C, name="lt_set_rng_window_static") result(rc)
      import :: c_ptr, c_int, c_int64_t
      type(c_ptr), value :: engine
      integer(c_int64_t), value :: active_before, active_total
      integer(c_int) :: rc
    end function
    ! per-phase GPU time (WriteModelTiming): ms4 = set-up, advection, vertical turbulence, update
    function lt_phase_timing(engine, on) bind(C, name="lt_phase_timing") result(rc)
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: engine
      integer(c_int32_t), value :: on
      integer(c_int) :: rc
    end function
    function lt_get_phase_times(engine, ms4, reset) bind(C, name="lt_get_phase_times") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: engine
      real(c_double), intent(out) :: ms4(4)
      integer(c_int32_t), value :: reset
      integer(c_int) :: rc
    end function
    ! multi-GPU: the RCCL all-gather of the shards' (x,y,z) behind the ABI (one process per GPU)
    function lt_comm_init_file(engine, path, nranks, rank, timeout_s) bind(C, name="lt_comm_init_file") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_char
      type(c_ptr), value :: engine
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int32_t), value :: nranks, rank, timeout_s
      integer(c_int) :: rc
    end function
    function lt_allgather_positions(engine, counts, xyz_host, xyz_device) bind(C, name="lt_allgather_positions") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: engine
      integer(c_int32_t), intent(in) :: counts(*)
      real(c_double), intent(out) :: xyz_host(*)
      type(c_ptr), value :: xyz_device
      integer(c_int) :: rc
    end function
    function lt_allgather_wait(engine) bind(C, name="lt_allgather_wait") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: engine
      integer(c_int) :: rc
    end function
    function lt_comm_destroy(engine) bind(C, name="lt_comm_destroy") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: engine
      integer(c_int) :: rc
    end function
    function lt_step(engine, ex, ix, p, it) bind(C, name="lt_step") result(rc)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), value :: engine
      real(c_double), intent(in) :: ex(3), ix(3)
      integer(c_int32_t), value :: p, it
      integer(c_int) :: rc
    end function
    function lt_sync(engine) bind(C, name="lt_sync") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: engine
      integer(c_int) :: rc
    end function
    function lt_get_stats(engine, stats) bind(C, name="lt_get_stats") result(rc)
      import :: c_int, c_ptr, lt_step_stats
      type(c_ptr), value :: engine
      type(lt_step_stats), intent(out) :: stats
      integer(c_int) :: rc
    end function
    function lt_pack_positions(engine, dst_device) bind(C, name="lt_pack_positions") result(rc)
      import :: c_int, c_ptr
      type(c_ptr), value :: engine, dst_device
      integer(c_int) :: rc
    end function
    ! ---- init-time geometry on the device (SURVEY 8f row 4) ----
    ! the adjacency loops of initGrid (hydrodynamic_module.f90:588-645); adjacent(elements,10)
    function lt_build_adjacency(device, elements, row, E, adjacent) bind(C, name="lt_build_adjacency") result(rc)
      import :: c_int, c_int32_t
      integer(c_int32_t), value :: device, elements, row
      integer(c_int32_t), intent(in) :: E(4, *)
      integer(c_int32_t), intent(out) :: adjacent(elements, 10)
      integer(c_int) :: rc
    end function
    ! setEle_all (hydrodynamic_module.f90:1536-1676)
    function lt_locate_particles(engine, n, Xpar, Ypar, P_r_element, P_u_element, P_v_element, err, par) &
        bind(C, name="lt_locate_particles") result(rc)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), value :: engine
      integer(c_int32_t), value :: n
      real(c_double), intent(in) :: Xpar(*), Ypar(*)
      integer(c_int32_t), intent(out) :: P_r_element(*), P_u_element(*), P_v_element(*)
      integer(c_int32_t), intent(out) :: err, par
      integer(c_int) :: rc
    end function
    ! createPolySpecs (settlement_module.f90:245-480); the list members of h may be c_null_ptr
    ! createBounds (boundary_module.f90:82-1215) on the host side of the library
    function lt_create_bounds(vi, uj, mask_rho, x_u, y_u, x_v, y_v, cap, nbounds, maxbound, maxisland, numislands, &
                              bx, by, hx, hy, hid, bnd_x, bnd_y, land) bind(C, name="lt_create_bounds") result(rc)
      import :: c_int, c_int32_t, c_double
      integer(c_int32_t), value :: vi, uj, cap
      integer(c_int32_t), intent(in) :: mask_rho(*)
      real(c_double), intent(in) :: x_u(*), y_u(*), x_v(*), y_v(*)
      integer(c_int32_t), intent(out) :: nbounds, maxbound, maxisland, numislands
      real(c_double), intent(out) :: bx(*), by(*), hx(*), hy(*), bnd_x(*), bnd_y(*)
      integer(c_int32_t), intent(out) :: hid(*), land(*)
      integer(c_int) :: rc
    end function
    function lt_create_polyspecs(engine, h) bind(C, name="lt_create_polyspecs") result(rc)
      import :: c_int, c_ptr, lt_habitat
      type(c_ptr), value :: engine
      type(lt_habitat), intent(in) :: h
      integer(c_int) :: rc
    end function
    function lt_get_polyspecs(engine, polyspecs, holespecs, elepolys_ptr, elepolys_ids, elepolys_cap, &
                              polyholes_ptr, polyholes_ids, polyholes_cap) bind(C, name="lt_get_polyspecs") result(rc)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: engine
      type(c_ptr), value :: polyspecs, holespecs, elepolys_ptr, elepolys_ids, polyholes_ptr, polyholes_ids
      integer(c_int32_t), value :: elepolys_cap, polyholes_cap
      integer(c_int) :: rc
    end function
  end interface

contains

  ! lt_config with the sizes filled in and LTRANS.data defaults for the rest
  function lt_config_init() result(cfg)
    type(lt_config) :: cfg
    cfg%struct_size = int(c_sizeof(cfg), c_int32_t)
    cfg%abi_version = LT_ABI_EXPECTED
    cfg%numpar = 0; cfg%dt = 3600; cfg%idt = 120; cfg%us = 0; cfg%ws = 0
    cfg%hc = 0.2; cfg%Vtransform = 1; cfg%z0 = 0.0005d0
    cfg%HTurbOn = 0; cfg%VTurbOn = 0; cfg%ConstantHTurb = 1.0d0
    cfg%Behavior = 0; cfg%OpenOceanBoundary = 0; cfg%mortality = 0
    cfg%deadage = 367200.d0; cfg%pediage = 302400.d0; cfg%swimstart = 0.d0
    cfg%swimslow = 0.005d0; cfg%swimfast = 0.005d0; cfg%sink = -0.0003d0; cfg%Swimdepth = 2.d0
    cfg%Sgradient = 1.0d0; cfg%Hswimspeed = 0.9d0
    cfg%twistart = 4.801821d0; cfg%twiend = 19.19956d0; cfg%daylength = 14.39774d0
    cfg%Em = 1814.328d0; cfg%Kd = 1.07d0; cfg%thresh = 0.0166d0
    cfg%settlementon = 0; cfg%holesExist = 0
    cfg%PI = 3.14159265358979d0
    cfg%Earth_Radius = 6378000.d0; cfg%SphericalProjection = 0; cfg%lonmin = -77.d0; cfg%latmin = 36.d0
    cfg%seed = 9; cfg%ErrorFlag = 1; cfg%SaltTempOn = 0; cfg%TrackCollisions = 0; cfg%FreeSlip = 0
    cfg%slab_storage = LT_SLAB_F32; cfg%device = -1; cfg%resort_every = 0
  end function

  ! engine-owned message of the last failure as a Fortran string
  function lt_error_string(engine) result(s)
    type(c_ptr), intent(in) :: engine
    character(len=:), allocatable :: s
    character(kind=c_char), pointer :: p(:)
    type(c_ptr) :: cp
    integer :: n
    cp = lt_last_error(engine)
    s = ''
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, p, [512])
    n = 0
    do while (n < 512)
      if (p(n + 1) == c_null_char) exit
      n = n + 1
    enddo
    allocate(character(len=n) :: s)
    if (n > 0) s = transfer(p(1:n), s)
  end function

  subroutine lt_check(rc, engine, what)
    integer(c_int), intent(in) :: rc
    type(c_ptr), intent(in) :: engine
    character(len=*), intent(in) :: what
    if (rc < 0) then
      write(*,*) 'ltrans_hip: ', what, ' failed with status ', rc
      if (c_associated(engine)) write(*,*) '  ', lt_error_string(engine)
      write(*,*) 'The Program Cannot Continue and Will Terminate'
      stop 1
    endif
  end subroutine

end module ltrans_hip
