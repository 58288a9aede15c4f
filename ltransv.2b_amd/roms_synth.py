"""Synthetic ROMS grid / history / particles and the init-time host geometry.

The reference reads a ROMS grid + history from NetCDF (hydrodynamic_module.f90:
88-1049) and derives, once, on the host, everything the particle loop later
reads: node-numbered coordinates, wet-element lists RE/UE/VE, 10-wide adjacency
lists (:590-647), boundary polylines (createBounds, boundary_module.f90:82-1215)
and habitat polygon indexes (settlement_module.f90:245-480).  Those steps are
host-side and out of the hot path; this module provides
deterministic stand-ins so that tests and bench.py can drive the engine on the
shapes BASELINE.json names (SURVEY.md section 8d "Synthetic inputs").  The
outputs use the reference's conventions exactly (1-based node/element ids,
node = i + (j-1)*vi, element corner order (i,j),(i+1,j),(i+1,j+1),(i,j+1),
adjacency = self first then neighbours in ascending id, zero filled).
"""
import math

import numpy as np


# --------------------------------------------------------------------------
# grid
# --------------------------------------------------------------------------
def s_coordinates(us, theta_s=5.0, theta_b=0.4):
    """ROMS Song-Haidvogel stretching: (SC, CS) on rho levels, (SCW, CSW) on w levels."""
    def cs(sc):
        return ((1.0 - theta_b) * np.sinh(theta_s * sc) / math.sinh(theta_s)
                + theta_b * (np.tanh(theta_s * (sc + 0.5)) / (2.0 * math.tanh(0.5 * theta_s)) - 0.5))
    k = np.arange(1, us + 1, dtype=np.float64)
    sc = (k - us - 0.5) / us
    kw = np.arange(0, us + 1, dtype=np.float64)
    scw = (kw - us) / us
    csw = cs(scw)
    csw[0], csw[-1] = -1.0, 0.0
    return sc, cs(sc), scw, csw


def make_grid(vi, uj, us, dx=1000.0, dy=1000.0, curvilinear=0.08, islands=(), open_east=False,
              hmin=20.0, hmax=200.0):
    """Rectilinear grid with a smooth curvilinear perturbation, closed land rim,
    optional rectangular islands ``(i0, i1, j0, j1)`` (rho indices, inclusive,
    0-based) and optionally an open (water) eastern rim."""
    ui, vj = vi - 1, uj - 1
    i = np.arange(vi, dtype=np.float64)[:, None]
    j = np.arange(uj, dtype=np.float64)[None, :]
    px = curvilinear * dx * np.sin(2 * np.pi * 1.5 * i / vi + 0.3) * np.cos(2 * np.pi * 1.0 * j / uj + 0.2)
    py = curvilinear * dy * np.cos(2 * np.pi * 1.0 * i / vi - 0.4) * np.sin(2 * np.pi * 1.5 * j / uj + 0.1)
    x_rho = i * dx + px + 0.0 * j
    y_rho = j * dy + py + 0.0 * i
    x_u = 0.5 * (x_rho[:-1, :] + x_rho[1:, :])
    y_u = 0.5 * (y_rho[:-1, :] + y_rho[1:, :])
    x_v = 0.5 * (x_rho[:, :-1] + x_rho[:, 1:])
    y_v = 0.5 * (y_rho[:, :-1] + y_rho[:, 1:])
    mid = 0.5 * (hmin + hmax)
    amp = 0.5 * (hmax - hmin)
    h = mid + amp * np.sin(2 * np.pi * i / vi * 1.25) * np.cos(2 * np.pi * j / uj * 0.75)
    angle = 0.1 * np.sin(2 * np.pi * i / vi) * np.sin(2 * np.pi * j / uj + 0.5)
    mask = np.ones((vi, uj), dtype=np.int32)
    mask[0, :] = 0
    mask[:, 0] = 0
    mask[:, -1] = 0
    if not open_east:
        mask[-1, :] = 0
    for (i0, i1, j0, j1) in islands:
        mask[i0:i1 + 1, j0:j1 + 1] = 0
    sc, cs, scw, csw = s_coordinates(us)
    return dict(vi=vi, uj=uj, ui=ui, vj=vj, us=us, ws=us + 1, dx=dx, dy=dy,
                x_rho=x_rho, y_rho=y_rho, x_u=x_u, y_u=y_u, x_v=x_v, y_v=y_v,
                h=h, angle=angle, mask_rho=mask, SC=sc, CS=cs, SCW=scw, CSW=csw,
                open_east=bool(open_east))


def _compact_elements(ni, nj, node_mask_flat):
    """Elements of an (ni x nj) node lattice with >=1 water node.
    Returns (E (4,n) int32 1-based node ids, compact id map (ni-1, nj-1) 1-based / 0)."""
    ei = np.arange(ni - 1)[:, None]
    ej = np.arange(nj - 1)[None, :]
    n1 = (ei + 1) + ej * ni
    nodes = np.stack([n1, n1 + 1, n1 + 1 + ni, n1 + ni], axis=0)          # (4, ni-1, nj-1)
    wet = np.zeros((ni - 1, nj - 1), dtype=bool)
    for k in range(4):
        wet |= node_mask_flat[nodes[k] - 1] == 1
    order = np.flatnonzero(wet.ravel(order="F"))                            # i fastest
    cmap = np.zeros((ni - 1) * (nj - 1), dtype=np.int32)
    cmap[order] = np.arange(1, order.size + 1, dtype=np.int32)
    E = nodes.reshape(4, -1, order="F")[:, order].astype(np.int32)
    return np.asfortranarray(E), cmap.reshape(ni - 1, nj - 1, order="F")


def _adjacency(cmap):
    """(elements,10): self, then node-sharing neighbours ascending, zero filled
    (hydrodynamic_module.f90:590-647)."""
    ni, nj = cmap.shape
    pad = np.zeros((ni + 2, nj + 2), dtype=np.int32)
    pad[1:-1, 1:-1] = cmap
    neigh = []
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            if di == 0 and dj == 0:
                continue
            neigh.append(pad[1 + di:ni + 1 + di, 1 + dj:nj + 1 + dj])
    neigh = np.stack(neigh, axis=-1).reshape(-1, 8, order="F")            # full-lattice order
    flat = cmap.ravel(order="F")
    sel = flat > 0
    nb = neigh[sel]
    big = np.iinfo(np.int32).max
    nb = np.where(nb == 0, big, nb)
    nb.sort(axis=1)
    nb = np.where(nb == big, 0, nb)
    adj = np.zeros((nb.shape[0], 10), dtype=np.int32)
    adj[:, 0] = flat[sel]
    adj[:, 1:9] = nb
    return np.asfortranarray(adj)


def build_geometry(g):
    """initGrid equivalent: everything lt_set_grid needs."""
    vi, uj, ui, vj = g["vi"], g["uj"], g["ui"], g["vj"]
    F = lambda a: np.ascontiguousarray(a.ravel(order="F"))                  # noqa: E731
    rho_mask = F(g["mask_rho"]).astype(np.int32)
    m = g["mask_rho"]
    mask_u = (m[:-1, :] * m[1:, :]).astype(np.int32)
    mask_v = (m[:, :-1] * m[:, 1:]).astype(np.int32)
    u_mask, v_mask = F(mask_u), F(mask_v)
    RE, rmap = _compact_elements(vi, uj, rho_mask)
    UE, umap = _compact_elements(ui, uj, u_mask)
    VE, vmap = _compact_elements(vi, vj, v_mask)
    out = dict(
        rho_nodes=vi * uj, u_nodes=ui * uj, v_nodes=vi * vj,
        rho_elements=RE.shape[1], u_elements=UE.shape[1], v_elements=VE.shape[1],
        rx=F(g["x_rho"]), ry=F(g["y_rho"]), ux=F(g["x_u"]), uy=F(g["y_u"]),
        vx=F(g["x_v"]), vy=F(g["y_v"]), depth=F(g["h"]), rho_angle=F(g["angle"]),
        rho_mask=rho_mask, u_mask=u_mask, v_mask=v_mask,
        RE=RE, UE=UE, VE=VE,
        r_Adjacent=_adjacency(rmap), u_Adjacent=_adjacency(umap), v_Adjacent=_adjacency(vmap),
        SC=g["SC"], CS=g["CS"], SCW=g["SCW"], CSW=g["CSW"],
        _rmap=rmap, _umap=umap, _vmap=vmap,
        m_r=m.astype(np.float64), m_u=mask_u.astype(np.float64), m_v=mask_v.astype(np.float64),
    )
    return out


# --------------------------------------------------------------------------
# boundaries (createBounds equivalent)
# --------------------------------------------------------------------------
def create_bounds(g):
    """Boundary polylines through the U/V points that separate water from land
    (or from the outermost rho ring), as createBounds produces them: closed main
    polygon bx/by, closed island polygons hx/hy/hid (ids from 1001), the segment
    table bnd_x/bnd_y(2,nbounds) and land(nbounds) (.FALSE. on open ocean)."""
    vi, uj = g["vi"], g["uj"]
    m = g["mask_rho"]
    inside = np.zeros((vi, uj), dtype=bool)
    inside[1:-1, 1:-1] = m[1:-1, 1:-1] == 1
    # boundary points keyed ('u', i, j) between rho(i,j),(i+1,j); ('v', i, j) between (i,j),(i,j+1)
    links = {}

    def add(a, b):
        links.setdefault(a, []).append(b)
        links.setdefault(b, []).append(a)

    for j in range(uj - 1):
        for i in range(vi - 1):
            c00, c10, c01, c11 = inside[i, j], inside[i + 1, j], inside[i, j + 1], inside[i + 1, j + 1]
            pts = []
            if c00 != c10:
                pts.append(("u", i, j))
            if c10 != c11:
                pts.append(("v", i + 1, j))
            if c01 != c11:
                pts.append(("u", i, j + 1))
            if c00 != c01:
                pts.append(("v", i, j))
            if len(pts) == 2:
                add(pts[0], pts[1])
            elif len(pts) == 4:
                raise ValueError("ambiguous (checkerboard) mask block at %d,%d" % (i, j))
    loops, seen = [], set()
    for start in sorted(links):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxts = [q for q in links[cur] if q != prev]
            if len(links[cur]) != 2:
                raise ValueError("open boundary chain at %r" % (cur,))
            nxt = nxts[0] if nxts else links[cur][0]
            if nxt == start:
                break
            loop.append(nxt)
            seen.add(nxt)
            prev, cur = cur, nxt
        loops.append(loop)

    def xy(pt):
        kind, i, j = pt
        return (g["x_u"][i, j], g["y_u"][i, j]) if kind == "u" else (g["x_v"][i, j], g["y_v"][i, j])

    def is_oo(pt):
        kind, i, j = pt
        if kind == "u":
            return (i == 0 and m[0, j] == 1) or (i == vi - 2 and m[vi - 1, j] == 1)
        return (j == 0 and m[i, 0] == 1) or (j == uj - 2 and m[i, uj - 1] == 1)

    def area(loop):
        p = np.array([xy(q) for q in loop])
        return 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))

    loops.sort(key=area, reverse=True)
    main, isl = loops[0], loops[1:]
    mainc = main + [main[0]]
    bx = np.array([xy(q)[0] for q in mainc])
    by = np.array([xy(q)[1] for q in mainc])
    oo = np.array([is_oo(q) for q in mainc])
    hx, hy, hid = [], [], []
    for k, loop in enumerate(isl):
        for q in loop + [loop[0]]:
            hx.append(xy(q)[0])
            hy.append(xy(q)[1])
            hid.append(1001 + k)
    hx, hy, hid = np.array(hx, dtype=np.float64), np.array(hy, dtype=np.float64), np.array(hid, dtype=np.int32)
    maxbound, maxisland, numislands = bx.size, hx.size, len(isl)
    nbounds = maxbound + maxisland - numislands - 1
    bnd_x = np.zeros((2, nbounds))
    bnd_y = np.zeros((2, nbounds))
    land = np.ones(nbounds, dtype=np.int32)
    for i in range(maxbound - 1):
        bnd_x[:, i] = (bx[i], bx[i + 1])
        bnd_y[:, i] = (by[i], by[i + 1])
        if oo[i] and oo[i + 1]:
            land[i] = 0
    if numislands > 0:                                   # boundary_module.f90:1186-1201
        jj, kk = 1, hid[0]
        for i in range(1, maxisland):
            if hid[i - 1] != kk:
                kk = hid[i - 1]
                jj += 1
            idx = maxbound + i - jj - 1
            bnd_x[:, idx] = (hx[i - 1], hx[i])
            bnd_y[:, idx] = (hy[i - 1], hy[i])
    return dict(nbounds=nbounds, maxbound=maxbound, maxisland=maxisland, numislands=numislands,
                bnd_x=np.asfortranarray(bnd_x), bnd_y=np.asfortranarray(bnd_y), land=land,
                bx=bx, by=by, hx=hx, hy=hy, hid=hid)


# --------------------------------------------------------------------------
# fields
# --------------------------------------------------------------------------
def make_record(g, geo, r, dt=3600.0, with_kh=True, drift=(0.0, 0.0), raw=None, with_st=True):
    """History record r (time r*dt) as ROMS stores it (float32), then masked and
    node-numbered as updateHydro does (hydrodynamic_module.f90:1371-1403).
    ``drift`` adds a uniform (u, v) to push particles into the coast (reflection tests).
    Returns dict zeta (rho_nodes), Uvel (u_nodes,us), Vvel (v_nodes,us),
    Wvel/Kh (rho_nodes,ws), all float64 holding float32-representable values.
    ``raw`` = np.float32 / np.float64: instead return the record as NF90_GET_VAR delivers
    it, BEFORE the mask multiply: dict romZ (vi,uj), romU (ui,uj,us), romV (vi,vj,us),
    romW, romKH (vi,uj,ws) of that type, Fortran order (land points hold ocean-like
    values, as a ROMS file's fill-free fields may).  with_st=False leaves salinity and temperature
    out (None): two transcendental fields over the whole grid that only the salinity / temperature
    behaviours read."""
    vi, uj, us, ws = g["vi"], g["uj"], g["us"], g["ws"]
    Lx, Ly = (vi - 1) * g["dx"], (uj - 1) * g["dy"]
    t = r * dt
    tide = 0.6 + 0.4 * math.cos(2 * math.pi * t / 44714.0)
    F = lambda a: a.reshape(-1, a.shape[-1], order="F") if a.ndim == 3 else a.ravel(order="F")  # noqa: E731

    def f32(a):
        return a.astype(np.float32).astype(np.float64)

    xr, yr = g["x_rho"] / Lx, g["y_rho"] / Ly
    zeta = 0.5 * np.sin(2 * np.pi * (xr - t / 44714.0)) * np.cos(np.pi * yr)
    sr = g["SC"][None, None, :]
    sw = g["SCW"][None, None, :]
    shear = ((sr + 1.0) * 0.9 + 0.1) ** 0.25
    xu, yu = g["x_u"] / Lx, g["y_u"] / Ly
    xv, yv = g["x_v"] / Lx, g["y_v"] / Ly
    # double gyre stream function psi = sin(2 pi x) sin(pi y): u = -dpsi/dy, v = dpsi/dx
    u2 = -0.5 * tide * np.sin(2 * np.pi * xu) * np.cos(np.pi * yu) + 0.05 * math.sin(2 * math.pi * t / 44714.0)
    v2 = 0.5 * tide * np.cos(2 * np.pi * xv) * np.sin(np.pi * yv) * (Ly / Lx) * 2.0
    v2 = np.clip(v2, -0.5, 0.5)
    # The 3-D fields level by level from their 2-D factors (each element through the same operations, in the same
    # order, as the whole-array expressions kept in the comments: the values are those of the earlier rounds bit for bit)
    u2d, v2d = u2 + drift[0], v2 + drift[1]
    shear_l = shear[0, 0, :]
    A2 = np.sin(2 * np.pi * xr) * np.cos(np.pi * yr)
    cw_l = 1.0e-3 * np.sin(np.pi * g["SCW"])
    U_l = lambda l: u2d * shear_l[l]            # noqa: E731  U = (u2[:, :, None] + drift[0]) * shear
    V_l = lambda l: v2d * shear_l[l]            # noqa: E731  V = (v2[:, :, None] + drift[1]) * shear
    W_l = lambda l: cw_l[l] * A2 * tide         # noqa: E731  W = 1.0e-3 * sin(pi * sw) * A2[:, :, None] * tide
    levels = {"U": (U_l, us), "V": (V_l, us), "W": (W_l, ws)}
    if with_kh:
        B2 = 0.55 + 0.45 * np.sin(2 * np.pi * (xr + 0.25 * t / 44714.0)) * np.cos(2 * np.pi * yr)
        ck_l = 1.0e-2 * 4.0 * (-g["SCW"]) * (1.0 + g["SCW"])
        levels["K"] = (lambda l: 1.0e-5 + ck_l[l] * B2, ws)   # K = 1.0e-5 + 1.0e-2 * 4.0 * (-sw) * (1.0 + sw) * B2[:, :, None]
    # salinity with a halocline near -6 m that moves with the tide (up to ~2 psu/m: above the
    # Sgradient cue of behaviours 4/5 in places, below it elsewhere); temperature
    if with_st:
        hal2 = -6.0 + 2.0 * np.sin(2 * np.pi * (xr + t / 44714.0))
        C2 = 1.5 * (np.sin(2 * np.pi * xr) * np.cos(np.pi * yr))
        D2 = 0.5 * np.cos(2 * np.pi * (yr - t / 44714.0))
        zr_l = lambda l: g["h"] * g["SC"][l]    # noqa: E731  approximate depth of the rho levels
        # S = 30.0 + 4.0 * tanh((hal - zr) / 2.0) + 1.5 * (...)[:, :, None];  T = 12.0 + 8.0 * exp(zr / 15.0) + 0.5 * cos(...)[:, :, None]
        levels["S"] = (lambda l: 30.0 + 4.0 * np.tanh((hal2 - zr_l(l)) / 2.0) + C2, us)
        levels["T"] = (lambda l: 12.0 + 8.0 * np.exp(zr_l(l) / 15.0) + D2, us)

    def whole(name):
        fn, nl = levels[name]
        return np.stack([fn(l) for l in range(nl)], axis=2)

    if raw is not None:
        conv = lambda a: np.asfortranarray(a.astype(np.float32).astype(raw))  # noqa: E731
        return dict(romZ=conv(zeta), romU=conv(whole("U") + 0.0 * sr), romV=conv(whole("V") + 0.0 * sr), romW=conv(whole("W")),
                    romKH=conv(whole("K")) if with_kh else None, romS=conv(whole("S")) if with_st else None,
                    romT=conv(whole("T")) if with_st else None)

    def node_numbered(name, m2):
        """f32(field) * mask as (nodes, levels), Fortran order, node = i + j * ni: formed level by level into the
        level-major buffer whose transposed view that is (no 3-D temporaries, no transposing copy); the product
        with the 0 / 1 mask is exact in either type, so it is taken on the float32 values."""
        fn, nl = levels[name]
        ni, nj = m2.shape
        buf = np.empty((nl, nj, ni), dtype=np.float64)
        m32 = np.ascontiguousarray(m2.T, dtype=np.float32)
        for l in range(nl):
            lev = np.ascontiguousarray(fn(l).T, dtype=np.float32)
            lev *= m32
            buf[l] = lev
        return buf.reshape(nl, nj * ni).T

    out = dict(
        zeta=F(f32(zeta) * geo["m_r"]),
        Uvel=node_numbered("U", geo["m_u"]),
        Vvel=node_numbered("V", geo["m_v"]),
        Wvel=node_numbered("W", geo["m_r"]),
        salt=node_numbered("S", geo["m_r"]) if with_st else None,
        temp=node_numbered("T", geo["m_r"]) if with_st else None,
        Kh=node_numbered("K", geo["m_r"]) if with_kh else None,
    )
    assert out["Uvel"].shape == (geo["u_nodes"], us) and out["Wvel"].shape == (geo["rho_nodes"], ws)
    return out


# --------------------------------------------------------------------------
# particles
# --------------------------------------------------------------------------
def _in_quad(px, py, qx, qy):
    """vectorised point-in-convex-quad (corners counter-clockwise), edges inclusive."""
    ok = np.ones(px.shape, dtype=bool)
    for a in range(4):
        b = (a + 1) % 4
        cr = (qx[b] - qx[a]) * (py - qy[a]) - (qy[b] - qy[a]) * (px - qx[a])
        ok &= cr >= 0.0
    return ok


def locate_elements(g, geo, x, y, kind):
    """setEle_all stand-in (hydrodynamic_module.f90:1536-1676): element id (1-based,
    compact numbering) of each point on the rho/u/v grid; 0 if none."""
    if kind == "r":
        X, Y, cmap, x0, y0 = g["x_rho"], g["y_rho"], geo["_rmap"], 0.0, 0.0
    elif kind == "u":
        X, Y, cmap, x0, y0 = g["x_u"], g["y_u"], geo["_umap"], 0.5 * g["dx"], 0.0
    else:
        X, Y, cmap, x0, y0 = g["x_v"], g["y_v"], geo["_vmap"], 0.0, 0.5 * g["dy"]
    ni, nj = cmap.shape
    gi = np.clip(np.floor((x - x0) / g["dx"]).astype(np.int64), 0, ni - 1)
    gj = np.clip(np.floor((y - y0) / g["dy"]).astype(np.int64), 0, nj - 1)
    ele = np.zeros(x.shape, dtype=np.int32)
    todo = np.arange(x.size)          # the points not located yet: the first cell that holds a point wins
    for dj in (0, -1, 1, -2, 2):
        for di in (0, -1, 1, -2, 2):
            if todo.size == 0:
                return ele
            ci = np.clip(gi[todo] + di, 0, ni - 1)
            cj = np.clip(gj[todo] + dj, 0, nj - 1)
            qx = [X[ci, cj], X[ci + 1, cj], X[ci + 1, cj + 1], X[ci, cj + 1]]
            qy = [Y[ci, cj], Y[ci + 1, cj], Y[ci + 1, cj + 1], Y[ci, cj + 1]]
            cm = cmap[ci, cj]
            hit = _in_quad(x[todo], y[todo], qx, qy) & (cm > 0)
            ele[todo[hit]] = cm[hit]
            todo = todo[~hit]
    return ele


def make_particles(g, geo, bounds, numpar, seed=12345, margin_cells=3, near_land=0.0, dob=0.0,
                   coast_band=None):
    """Initial particles: uniform over the interior >= margin_cells from the rim
    and away from islands, depth uniform in (-0.9 h, -0.5 m); ``near_land`` =
    fraction seeded 0.7-2.5 cells from the west/south rim to force reflections
    (``coast_band=(lo, hi)``: that fraction lo..hi metres inside the west/south
    boundary line itself, so that a few steps of drift reach it).
    Returns the arrays of lt_set_particles (par(numpar,13), P_*_element) with the
    elements already located (setEle_all stand-in)."""
    rng = np.random.default_rng(seed)
    vi, uj, dx, dy = g["vi"], g["uj"], g["dx"], g["dy"]
    m = g["mask_rho"]
    rim = np.zeros_like(m, dtype=bool)
    rim[0, :] = rim[-1, :] = rim[:, 0] = rim[:, -1] = True
    x = np.empty(numpar)
    y = np.empty(numpar)
    n_near = int(round(near_land * numpar))
    todo = np.arange(numpar)
    rounds = 0
    while todo.size:
        k = todo.size
        near = todo < n_near
        mc = float(margin_cells)
        xs = (mc + rng.uniform(0.0, 1.0, k) * (vi - 1 - 2 * mc)) * dx
        ys = (mc + rng.uniform(0.0, 1.0, k) * (uj - 1 - 2 * mc)) * dy
        side = rng.integers(0, 2, k)
        band = 0.7 + 1.8 * rng.uniform(0.0, 1.0, k)
        if coast_band is None:
            xs = np.where(near & (side == 0), band * dx, xs)
            ys = np.where(near & (side == 1), band * dy, ys)
        else:
            # the boundary runs through the u points (0, j) in the west and the v points (i, 0)
            # in the south (create_bounds): interpolate it at the particle and step inside
            off = coast_band[0] + (coast_band[1] - coast_band[0]) * rng.uniform(0.0, 1.0, k)
            fj = np.clip(ys / dy, mc, uj - 1 - mc)
            j0 = np.floor(fj).astype(int)
            xb = g["x_u"][0, j0] + (fj - j0) * (g["x_u"][0, j0 + 1] - g["x_u"][0, j0])
            yw = g["y_u"][0, j0] + (fj - j0) * (g["y_u"][0, j0 + 1] - g["y_u"][0, j0])
            fi = np.clip(xs / dx, mc, vi - 1 - mc)
            i0 = np.floor(fi).astype(int)
            yb = g["y_v"][i0, 0] + (fi - i0) * (g["y_v"][i0 + 1, 0] - g["y_v"][i0, 0])
            xsouth = g["x_v"][i0, 0] + (fi - i0) * (g["x_v"][i0 + 1, 0] - g["x_v"][i0, 0])
            west, south = near & (side == 0), near & (side == 1)
            xs = np.where(west, xb + off, np.where(south, xsouth, xs))
            ys = np.where(west, yw, np.where(south, yb + off, ys))
        gi = np.clip(np.floor(xs / dx).astype(int), 0, vi - 2)
        gj = np.clip(np.floor(ys / dy).astype(int), 0, uj - 2)
        ok = np.ones(k, dtype=bool)
        for dj in (-1, 0, 1, 2):
            for di in (-1, 0, 1, 2):
                ii = np.clip(gi + di, 0, vi - 1)
                jj = np.clip(gj + dj, 0, uj - 1)
                inner = (di in (0, 1)) and (dj in (0, 1))
                water = m[ii, jj] == 1
                if inner:
                    ok &= water | (near & rim[ii, jj])
                else:
                    ok &= water | near | rim[ii, jj]
        x[todo[ok]] = xs[ok]
        y[todo[ok]] = ys[ok]
        todo = todo[~ok]
        rounds += 1
        if rounds > 500:
            raise RuntimeError("could not place particles")
    er = locate_elements(g, geo, x, y, "r")
    eu = locate_elements(g, geo, x, y, "u")
    ev = locate_elements(g, geo, x, y, "v")
    if (er == 0).any() or (eu == 0).any() or (ev == 0).any():
        raise RuntimeError("particle outside every element")
    gi = np.clip(np.floor(x / dx).astype(int), 0, vi - 2)
    gj = np.clip(np.floor(y / dy).astype(int), 0, uj - 2)
    hloc = np.minimum.reduce([g["h"][gi, gj], g["h"][gi + 1, gj], g["h"][gi, gj + 1],
                              g["h"][gi + 1, gj + 1]])
    z = -rng.uniform(0.5 / hloc, 0.9, numpar) * hloc
    par = np.zeros((numpar, 13), order="F")
    par[:, 0], par[:, 1], par[:, 2] = x, y, z
    par[:, 3:6] = par[:, 0:3]
    par[:, 6:9] = par[:, 0:3]
    par[:, 10] = dob
    return dict(par=par, P_r_element=er, P_u_element=eu, P_v_element=ev)


def behaviour_state(cfg, numpar):
    """initBehave / initSettlement equivalents (behavior_module.f90:94-160,
    settlement_module.f90:40-100)."""
    ped = np.full(numpar, cfg.pediage)
    swim = np.zeros((numpar, 3), order="F")
    swim[:, 0] = (cfg.swimfast - cfg.swimslow) / (ped - cfg.swimstart)
    swim[:, 1] = cfg.swimfast - swim[:, 0] * ped
    return dict(P_behave=np.full(numpar, cfg.Behavior, dtype=np.int32),
                dead=np.zeros(numpar, dtype=np.int32), oob=np.zeros(numpar, dtype=np.int32),
                P_swim=swim, P_pediage=ped, P_deadage=np.full(numpar, cfg.deadage),
                settle=np.zeros(numpar, dtype=np.int32), settletime=ped.copy(),
                endpoly=np.zeros(numpar, dtype=np.int32),
                hitBottom=np.zeros(numpar, dtype=np.int32), hitLand=np.zeros(numpar, dtype=np.int32),
                # behaviours 4, 5, 7: timer = 0, P_Sprev = P_zprev = 0, bottom = .TRUE. (:113,139-144)
                timer=np.zeros(numpar), P_Sprev=np.zeros(numpar), P_zprev=np.zeros(numpar),
                bottom=np.ones(numpar, dtype=np.int32))


# --------------------------------------------------------------------------
# habitat polygons (createPolySpecs equivalent)
# --------------------------------------------------------------------------
def make_habitat(g, geo, polygons, holes=(), minpolyid=101001, minholeid=100201):
    """polygons: list of (cx, cy, radius, nvert); holes: list of (poly_index, cx, cy, radius, nvert).
    Builds polys/holes tables, specs, max centre distances and the per-element /
    per-polygon id lists in the layout of settlement_module.f90:49-89."""
    rows, specs, maxb = [], [], []
    for k, (cx, cy, rad, nv) in enumerate(polygons):
        pid = minpolyid + k
        ang = 2 * np.pi * np.arange(nv + 1) / nv
        ex, ey = cx + rad * np.cos(ang), cy + rad * np.sin(ang)
        ex[-1], ey[-1] = ex[0], ey[0]
        specs.append((len(rows) + 1, nv + 1))
        for a, b in zip(ex, ey):
            rows.append((pid, cx, cy, a, b))
        maxb.append(np.sqrt((ex - cx) ** 2 + (ey - cy) ** 2).max())
    polys = np.asfortranarray(np.array(rows, dtype=np.float64).reshape(-1, 5))
    npoly = len(polygons)
    hrows, hspecs, maxh, ph = [], [], [], [[] for _ in range(npoly)]
    for k, (pidx, cx, cy, rad, nv) in enumerate(holes):
        hid = minholeid + k
        ang = 2 * np.pi * np.arange(nv + 1) / nv
        ex, ey = cx + rad * np.cos(ang), cy + rad * np.sin(ang)
        ex[-1], ey[-1] = ex[0], ey[0]
        hspecs.append((len(hrows) + 1, nv + 1))
        for a, b in zip(ex, ey):
            hrows.append((hid, cx, cy, a, b, minpolyid + pidx))
        maxh.append(np.sqrt((ex - cx) ** 2 + (ey - cy) ** 2).max())
        ph[pidx].append(hid)
    nhole = len(holes)
    holes_t = np.asfortranarray(np.array(hrows, dtype=np.float64).reshape(-1, 6)) if nhole else np.zeros((0, 6), order="F")
    # polygons per rho element: bbox overlap (a superset is harmless: psettle re-tests)
    RE = geo["RE"]
    exs = geo["rx"][RE - 1]
    eys = geo["ry"][RE - 1]
    exmin, exmax, eymin, eymax = exs.min(0), exs.max(0), eys.min(0), eys.max(0)
    lists = [[] for _ in range(RE.shape[1])]
    for k, (cx, cy, rad, nv) in enumerate(polygons):
        hit = np.flatnonzero((exmax >= cx - rad) & (exmin <= cx + rad) & (eymax >= cy - rad) & (eymin <= cy + rad))
        for e in hit:
            lists[e].append(minpolyid + k)
    ptr = np.zeros(RE.shape[1] + 1, dtype=np.int32)
    ptr[1:] = np.cumsum([len(v) for v in lists])
    ids = np.array([v for l in lists for v in l], dtype=np.int32)
    pptr = np.zeros(npoly + 1, dtype=np.int32)
    pptr[1:] = np.cumsum([len(v) for v in ph])
    pids = np.array([v for l in ph for v in l], dtype=np.int32)
    return dict(pedges=polys.shape[0], hedges=holes_t.shape[0],
                minpolyid=minpolyid, maxpolyid=minpolyid + npoly - 1,
                minholeid=minholeid, maxholeid=minholeid + max(nhole, 1) - 1,
                polys=polys, holes=holes_t,
                polyspecs=np.asfortranarray(np.array(specs, dtype=np.int32).reshape(-1, 2)),
                holespecs=np.asfortranarray(np.array(hspecs, dtype=np.int32).reshape(-1, 2)) if nhole
                else np.zeros((1, 2), dtype=np.int32, order="F"),
                maxbdis=np.array(maxb), maxhdis=np.array(maxh) if nhole else np.zeros(1),
                elepolys_ptr=ptr, elepolys_ids=ids if ids.size else np.zeros(1, dtype=np.int32),
                polyholes_ptr=pptr, polyholes_ids=pids if pids.size else np.zeros(1, dtype=np.int32))
