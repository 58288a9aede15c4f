"""ctypes mirror of include/ltrans_hip.h (ABI version 1).

Field order and types must match the header exactly; ``tests/test_abi.py``
checks sizes/offsets against a C probe compiled from the header.
"""
import ctypes as C

LT_ABI_VERSION = 2

LT_OK = 0
LT_ERR = {
    -1: "LT_ERR_ARG", -2: "LT_ERR_ABI", -3: "LT_ERR_STATE", -4: "LT_ERR_UNSUPPORTED",
    -5: "LT_ERR_DEVICE", -6: "LT_ERR_SLAB_NOT_F32", -7: "LT_ERR_NOMEM",
}
LT_SLAB_F32, LT_SLAB_F64 = 0, 1
LT_ERR_SLAB_NOT_F32 = -6

PERR_NAMES = {0: "none", 4: "not in rho element", 5: "not in u element", 6: "not in v element",
              24: "out after 3rd reflection", 25: "outside main bounds", 26: "inside island",
              27: "jumped over rho element", 28: "jumped over u element",
              29: "jumped over v element"}

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


class LtError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {LT_ERR.get(code, code)} {detail}".strip())


class LtConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("abi_version", C.c_int32),
        ("numpar", C.c_int32),
        ("dt", C.c_int32), ("idt", C.c_int32),
        ("us", C.c_int32), ("ws", C.c_int32),
        ("hc", C.c_float),
        ("Vtransform", C.c_int32),
        ("z0", C.c_double),
        ("HTurbOn", C.c_int32), ("VTurbOn", C.c_int32),
        ("ConstantHTurb", C.c_double),
        ("Behavior", C.c_int32), ("OpenOceanBoundary", C.c_int32), ("mortality", C.c_int32),
        ("deadage", C.c_double), ("pediage", C.c_double), ("swimstart", C.c_double),
        ("swimslow", C.c_double), ("swimfast", C.c_double), ("sink", C.c_double),
        ("Swimdepth", C.c_double),
        ("Sgradient", C.c_double), ("Hswimspeed", C.c_double),
        ("twistart", C.c_double), ("twiend", C.c_double), ("daylength", C.c_double),
        ("Em", C.c_double), ("Kd", C.c_double), ("thresh", C.c_double),
        ("settlementon", C.c_int32), ("holesExist", C.c_int32),
        ("PI", C.c_double), ("Earth_Radius", C.c_double), ("SphericalProjection", C.c_int32),
        ("lonmin", C.c_double), ("latmin", C.c_double),
        ("seed", C.c_int32), ("ErrorFlag", C.c_int32), ("SaltTempOn", C.c_int32),
        ("TrackCollisions", C.c_int32), ("FreeSlip", C.c_int32),
        ("slab_storage", C.c_int32), ("device", C.c_int32), ("resort_every", C.c_int32),
    ]

    def __init__(self, **kw):
        super().__init__()
        # LTRANS.data defaults (Model/LTRANS.data) for the keys the path reads
        self.struct_size = C.sizeof(LtConfig)
        self.abi_version = LT_ABI_VERSION
        self.dt, self.idt = 3600, 120
        self.hc, self.z0, self.Vtransform = 0.2, 0.0005, 1
        self.ConstantHTurb = 1.0
        self.deadage, self.pediage = 367200.0, 302400.0
        self.swimslow = self.swimfast = 0.005
        self.sink, self.Swimdepth = -0.0003, 2.0
        self.Sgradient, self.Hswimspeed = 1.0, 0.9
        self.twistart, self.twiend, self.daylength = 4.801821, 19.19956, 14.39774
        self.Em, self.Kd, self.thresh = 1814.328, 1.07, 0.0166
        self.PI = 3.14159265358979
        self.Earth_Radius, self.SphericalProjection = 6378000.0, 0
        self.lonmin, self.latmin = -77.0, 36.0
        self.seed = 9
        self.ErrorFlag = 1
        self.slab_storage = LT_SLAB_F32
        self.device = -1
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(f"lt_config has no field {k!r}")
            setattr(self, k, v)

    def copy(self, **kw):
        other = LtConfig()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(LtConfig))
        for k, v in kw.items():
            setattr(other, k, v)
        return other


class LtGrid(C.Structure):
    _fields_ = [
        ("rho_nodes", C.c_int32), ("u_nodes", C.c_int32), ("v_nodes", C.c_int32),
        ("rho_elements", C.c_int32), ("u_elements", C.c_int32), ("v_elements", C.c_int32),
        ("rx", c_dp), ("ry", c_dp), ("ux", c_dp), ("uy", c_dp), ("vx", c_dp), ("vy", c_dp),
        ("depth", c_dp), ("rho_angle", c_dp),
        ("rho_mask", c_ip), ("u_mask", c_ip), ("v_mask", c_ip),
        ("RE", c_ip), ("UE", c_ip), ("VE", c_ip),
        ("r_Adjacent", c_ip), ("u_Adjacent", c_ip), ("v_Adjacent", c_ip),
        ("SC", c_dp), ("CS", c_dp), ("SCW", c_dp), ("CSW", c_dp),
    ]


class LtBounds(C.Structure):
    _fields_ = [
        ("nbounds", C.c_int32), ("maxbound", C.c_int32), ("maxisland", C.c_int32),
        ("numislands", C.c_int32),
        ("bnd_x", c_dp), ("bnd_y", c_dp), ("land", c_ip),
        ("bx", c_dp), ("by", c_dp), ("hx", c_dp), ("hy", c_dp), ("hid", c_ip),
    ]


class LtHabitat(C.Structure):
    _fields_ = [
        ("pedges", C.c_int32), ("hedges", C.c_int32),
        ("minpolyid", C.c_int32), ("maxpolyid", C.c_int32),
        ("minholeid", C.c_int32), ("maxholeid", C.c_int32),
        ("polys", c_dp), ("holes", c_dp),
        ("polyspecs", c_ip), ("holespecs", c_ip),
        ("maxbdis", c_dp), ("maxhdis", c_dp),
        ("elepolys_ptr", c_ip), ("elepolys_ids", c_ip),
        ("polyholes_ptr", c_ip), ("polyholes_ids", c_ip),
    ]


PARTICLE_FIELDS = [
    # name, ctype pointer, numpy dtype, columns
    ("par", c_dp, "f8", 13),
    ("P_r_element", c_ip, "i4", 1), ("P_u_element", c_ip, "i4", 1), ("P_v_element", c_ip, "i4", 1),
    ("P_behave", c_ip, "i4", 1), ("dead", c_ip, "i4", 1), ("oob", c_ip, "i4", 1),
    ("P_swim", c_dp, "f8", 3), ("P_pediage", c_dp, "f8", 1), ("P_deadage", c_dp, "f8", 1),
    ("settle", c_ip, "i4", 1), ("settletime", c_dp, "f8", 1),
    ("endpoly", c_ip, "i4", 1), ("hitBottom", c_ip, "i4", 1), ("hitLand", c_ip, "i4", 1),
    ("err_code", c_ip, "i4", 1),
    ("timer", c_dp, "f8", 1), ("P_Sprev", c_dp, "f8", 1), ("P_zprev", c_dp, "f8", 1),
    ("bottom", c_ip, "i4", 1), ("P_Salt", c_dp, "f8", 1), ("P_Temp", c_dp, "f8", 1),
]


class LtParticles(C.Structure):
    _fields_ = [(name, ptr) for name, ptr, _, _ in PARTICLE_FIELDS]


class LtSlab(C.Structure):
    _fields_ = [("zeta", c_dp), ("Uvel", c_dp), ("Vvel", c_dp), ("Wvel", c_dp), ("Kh", c_dp),
                ("salt", c_dp), ("temp", c_dp)]


LT_ROM_F64, LT_ROM_F32 = 0, 1


class LtRomRecord(C.Structure):
    """lt_rom_record: a ROMS record as read from the history file (include/ltrans_hip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("dtype", C.c_int32),
        ("vi", C.c_int32), ("uj", C.c_int32), ("ui", C.c_int32), ("vj", C.c_int32),
        ("masked", C.c_int32), ("has_window", C.c_int32), ("window", C.c_int32 * 12),
        ("romZ", C.c_void_p), ("romU", C.c_void_p), ("romV", C.c_void_p), ("romW", C.c_void_p),
        ("romKH", C.c_void_p), ("romS", C.c_void_p), ("romT", C.c_void_p),
    ]


class LtOutput(C.Structure):
    _fields_ = [("lon", c_dp), ("lat", c_dp), ("z", c_dp), ("age", c_dp), ("status", c_dp),
                ("salt", c_dp), ("temp", c_dp), ("hitBottom", c_ip), ("hitLand", c_ip)]


class LtStepStats(C.Structure):
    _fields_ = [("active", C.c_int64), ("errors", C.c_int64), ("rng_draws", C.c_int64)]


# entry points every implementation of the ABI exports (prefix "lt_" for the HIP
# library; the test-side CPU oracle exports the same set as "lto_")
ENTRY_POINTS = [
    "create", "destroy", "set_grid", "set_bounds", "set_habitat", "set_particles",
    "set_slab", "set_time_slots", "slab_prefetch", "slab_commit", "set_rng", "get_rng", "count_active", "set_rng_window",
    "count_draws", "set_rng_window_draws",
    "step", "sync", "get_stats", "get_particles", "get_output",
    "build_adjacency", "locate_particles", "create_polyspecs", "get_polyspecs",
]
HIP_ONLY_ENTRY_POINTS = ["last_error", "abi_version", "source_hash", "set_rng_window_static", "pack_positions", "stream",
                         "phase_timing", "get_phase_times", "create_bounds", "comm_unique_id", "comm_init", "comm_init_file", "allgather_positions", "comm_destroy",
                         "exchange_rng_window",
                         "profile_begin", "profile_end", "get_counters"]
