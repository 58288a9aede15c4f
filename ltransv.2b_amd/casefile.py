"""Flat binary "case" file for the Fortran host harness (fortran/ltrans_hip_driver.f90)
and a writer for run files in the LTRANS.data namelist syntax.

NetCDF is not available in this image, so the arrays the reference derives from
its grid / history files (initGrid, createBounds, setEle_all, initHydro /
updateHydro) are handed to the Fortran harness in one stream file, in the fixed
order the harness reads them (all little-endian, Fortran column-major)."""
import numpy as np

MAGIC = 1280590641   # 'LTC1'
MAGIC2 = 1280590642  # 'LTC2': + the habitat polygons of initSettlement (settlement_module.f90:98-240)


def _f(a, dtype):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype).ravel(order="F")).tobytes()


def write_case(path, g, geo, bounds, parts, records, numpar, habitat=None):
    """records: list of dicts(zeta, Uvel, Vvel, Wvel, Kh) -- history records 0..n-1.
    habitat: the edge-point tables of the settlement polygons (polys(pedges,5), holes(hedges,6), the id ranges,
    maxbdis, maxhdis: what initSettlement reads from the habitat files); the harness builds the lists itself
    (lt_create_polyspecs = createPolySpecs)."""
    us, ws = g["us"], g["ws"]
    hdr = np.array([MAGIC if habitat is None else MAGIC2, g["vi"], g["uj"], us, ws, geo["rho_nodes"], geo["u_nodes"], geo["v_nodes"],
                    geo["rho_elements"], geo["u_elements"], geo["v_elements"], bounds["nbounds"],
                    bounds["maxbound"], bounds["maxisland"], bounds["numislands"], numpar,
                    len(records)], dtype=np.int32)
    with open(path, "wb") as fh:
        fh.write(hdr.tobytes())
        for k in ("rx", "ry", "ux", "uy", "vx", "vy", "depth", "rho_angle"):
            fh.write(_f(geo[k], np.float64))
        for k in ("RE", "UE", "VE", "r_Adjacent", "u_Adjacent", "v_Adjacent"):
            fh.write(_f(geo[k], np.int32))
        for k in ("SC", "CS", "SCW", "CSW"):
            fh.write(_f(geo[k], np.float64))
        fh.write(_f(bounds["bnd_x"], np.float64))
        fh.write(_f(bounds["bnd_y"], np.float64))
        fh.write(_f(bounds["land"], np.int32))
        fh.write(_f(bounds["bx"], np.float64))
        fh.write(_f(bounds["by"], np.float64))
        if bounds["maxisland"] > 0:
            fh.write(_f(bounds["hx"], np.float64))
            fh.write(_f(bounds["hy"], np.float64))
            fh.write(_f(bounds["hid"], np.int32))
        fh.write(_f(parts["par"], np.float64))
        for k in ("P_r_element", "P_u_element", "P_v_element"):
            fh.write(_f(parts[k], np.int32))
        if habitat is not None:
            h = habitat
            npoly = h["maxpolyid"] - h["minpolyid"] + 1
            nhole = (h["maxholeid"] - h["minholeid"] + 1) if h["hedges"] > 0 else 0
            fh.write(np.array([h["pedges"], h["hedges"], h["minpolyid"], h["maxpolyid"], h["minholeid"], h["maxholeid"]],
                              dtype=np.int32).tobytes())
            fh.write(_f(np.asarray(h["polys"]).reshape(h["pedges"], 5, order="F"), np.float64))
            if h["hedges"] > 0:
                fh.write(_f(np.asarray(h["holes"]).reshape(h["hedges"], 6, order="F"), np.float64))
            fh.write(_f(np.asarray(h["maxbdis"])[:npoly], np.float64))
            if nhole > 0:
                fh.write(_f(np.asarray(h["maxhdis"])[:nhole], np.float64))
        for r in records:
            kh = r.get("Kh")
            if kh is None:
                kh = np.zeros((geo["rho_nodes"], ws))
            for a in (r["zeta"], r["Uvel"], r["Vvel"], r["Wvel"], kh):
                fh.write(_f(a, np.float64))


def read_result(path):
    """Output of the harness: par(numpar,13) and the integer state."""
    with open(path, "rb") as fh:
        n, steps = np.frombuffer(fh.read(8), dtype=np.int32)
        par = np.frombuffer(fh.read(8 * 13 * n), dtype=np.float64).reshape(n, 13, order="F")
        out = {"par": par, "steps": int(steps)}
        for k in ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand",
                  "dead", "oob", "settle"):
            out[k] = np.frombuffer(fh.read(4 * n), dtype=np.int32)
        rest = fh.read()
        if len(rest) >= 8 * 5 * n:      # the print-step columns of lt_get_output
            cols = np.frombuffer(rest[:8 * 5 * n], dtype=np.float64).reshape(5, n)
            out.update(lon=cols[0], lat=cols[1], z=cols[2], age=cols[3], status=cols[4])
            rest = rest[8 * 5 * n:]
        if len(rest) >= 4:              # a sharded run: the gathered xyz(N,3) of all ranks
            ng = int(np.frombuffer(rest[:4], dtype=np.int32)[0])
            out["xyz_all"] = np.frombuffer(rest[4:4 + 8 * 3 * ng], dtype=np.float64).reshape(ng, 3, order="F")
    return out


def _l(v):
    return ".TRUE." if v else ".FALSE."


def write_ltrans_data(path, cfg, days=1.0, tdim=8, iprint=None, model_timing=False, habitat=None):
    """A run file in the LTRANS.data syntax with the groups the path reads."""
    hb = habitat or dict(minpolyid=101001, maxpolyid=101004, minholeid=100201, maxholeid=100401, pedges=1, hedges=1)
    txt = f"""! written by ltrans_amd.casefile.write_ltrans_data (LTRANS.data syntax)
$numparticles
  numpar = {cfg.numpar}          ! Number of particles
$end
$timeparam
  days   = {days}
  iprint = {iprint or cfg.dt}
  dt     = {cfg.dt}
  idt    = {cfg.idt}
$end
$hydroparam
  us         = {cfg.us}
  ws         = {cfg.ws}
  tdim       = {tdim}
  hc         = {float(np.float32(cfg.hc)):.9g}
  z0         = {cfg.z0!r}
  Vtransform = {cfg.Vtransform}
  readZeta   = .TRUE.
  constZeta  = 0.0
  readSalt   = .FALSE.
  constSalt  = 0.0
  readTemp   = .FALSE.
  constTemp  = 0.0
  readU      = .TRUE.
  constU     = 0.0
  readV      = .TRUE.
  constV     = 0.0
  readW      = .TRUE.
  constW     = 0.0
  readAks    = .TRUE.
  constAks   = 0.0
  readDens   = .FALSE.
  constDens  = 0.0
$end
$turbparam
  HTurbOn       = {_l(cfg.HTurbOn)}
  VTurbOn       = {_l(cfg.VTurbOn)}
  ConstantHTurb = {cfg.ConstantHTurb!r}
$end
$behavparam
  Behavior = {cfg.Behavior}
  OpenOceanBoundary = {_l(cfg.OpenOceanBoundary)}
  mortality = {_l(cfg.mortality)}
  deadage = {cfg.deadage!r}
  pediage = {cfg.pediage!r}
  swimstart = {cfg.swimstart!r}
  swimslow  = {cfg.swimslow!r}
  swimfast  = {cfg.swimfast!r}
  Sgradient = {cfg.Sgradient!r}
  sink      = {cfg.sink!r}
  Hswimspeed = {cfg.Hswimspeed!r}
  Swimdepth  = {cfg.Swimdepth!r}
$end
$dvmparam
  twistart  = {cfg.twistart!r}
  twiend    = {cfg.twiend!r}
  daylength = {cfg.daylength!r}
  Em        = {cfg.Em!r}
  Kd        = {cfg.Kd!r}
  thresh    = {cfg.thresh!r}
$end
$settleparam
 settlementon = {_l(cfg.settlementon)}
 holesExist = {_l(cfg.holesExist)}
 minpolyid  = {hb["minpolyid"]}
 maxpolyid  = {hb["maxpolyid"]}
 minholeid  = {hb["minholeid"]}
 maxholeid  = {hb["maxholeid"]}
 pedges     = {hb["pedges"]}
 hedges     = {hb["hedges"]}
$end
$convparam
  PI = {cfg.PI!r}
  Earth_Radius = {cfg.Earth_Radius!r}
  SphericalProjection = {_l(cfg.SphericalProjection)}
  latmin = {cfg.latmin!r}
  lonmin = {cfg.lonmin!r}
$end
$other
  seed         = {cfg.seed}
  ErrorFlag    = {cfg.ErrorFlag}
  BoundaryBLNs = .FALSE.
  SaltTempOn   = {_l(cfg.SaltTempOn)}
  TrackCollisions  = {_l(cfg.TrackCollisions)}
  WriteHeaders     = .FALSE.
  WriteModelTiming = {_l(model_timing)}
  ijbuff = 4
  FreeSlip = {_l(cfg.FreeSlip)}
$end
"""
    with open(path, "w") as fh:
        fh.write(txt)
