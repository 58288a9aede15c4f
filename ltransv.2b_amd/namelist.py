"""Reader for LTRANS.data: Fortran namelists in the ``$group ... $end`` form
(Model/LTRANS.data; schema Model/LTRANS.h:32-269, read in fixed order by
getParams parameter_module.f90:29-86).  Only the syntax the reference file uses
is accepted: ``name = value`` pairs, ``!`` comments, logicals ``.TRUE./.FALSE.``,
quoted strings, integers and reals.
"""
import re

from .abi import LtConfig

_GROUP = re.compile(r"^\s*[$&]\s*(\w+)\s*$")
_END = re.compile(r"^\s*([$&]\s*end|/)\s*$", re.I)


def _strip_comment(line):
    out, q = [], None
    for ch in line:
        if q:
            out.append(ch)
            if ch == q:
                q = None
        elif ch in "'\"":
            q = ch
            out.append(ch)
        elif ch == "!":
            break
        else:
            out.append(ch)
    return "".join(out)


def _value(tok):
    t = tok.strip().rstrip(",").strip()
    if not t:
        return None
    if t[0] in "'\"":
        return t[1:-1] if t[-1] == t[0] else t[1:]
    u = t.upper()
    if u in (".TRUE.", ".T.", "T"):
        return True
    if u in (".FALSE.", ".F.", "F"):
        return False
    try:
        return int(t)
    except ValueError:
        pass
    return float(u.replace("D", "E"))


def read_namelists(path):
    """-> {group(lower): {name(lower as written): value}}"""
    groups, cur = {}, None
    with open(path, "r", errors="replace") as fh:
        for raw in fh:
            line = _strip_comment(raw).strip()
            if not line:
                continue
            if _END.match(line):
                cur = None
                continue
            m = _GROUP.match(line)
            if m:
                cur = groups.setdefault(m.group(1).lower(), {})
                continue
            if cur is None:
                continue
            for part in re.split(r",(?=\s*\w+\s*=)", line):
                if "=" not in part:
                    continue
                k, v = part.split("=", 1)
                cur[k.strip().lower()] = _value(v)
    return groups


# namelist group -> (lt_config field, namelist key)
_MAP = {
    "numparticles": ["numpar"],
    "timeparam": ["dt", "idt"],
    "hydroparam": ["us", "ws", "hc", "z0", "Vtransform"],
    "turbparam": ["HTurbOn", "VTurbOn", "ConstantHTurb"],
    "behavparam": ["Behavior", "OpenOceanBoundary", "mortality", "deadage", "pediage",
                   "swimstart", "swimslow", "swimfast", "sink", "Swimdepth", "Sgradient", "Hswimspeed"],
    "dvmparam": ["twistart", "twiend", "daylength", "Em", "Kd", "thresh"],
    "settleparam": ["settlementon", "holesExist"],
    "convparam": ["PI", "Earth_Radius", "SphericalProjection", "lonmin", "latmin"],
    "other": ["seed", "ErrorFlag", "SaltTempOn", "TrackCollisions", "FreeSlip"],
}


def config_from_namelists(groups, **override):
    """Fill an lt_config from parsed LTRANS.data groups (hot-path keys only)."""
    cfg = LtConfig()
    for grp, fields in _MAP.items():
        vals = groups.get(grp, {})
        for f in fields:
            if f.lower() in vals and vals[f.lower()] is not None:
                v = vals[f.lower()]
                setattr(cfg, f, int(v) if isinstance(v, bool) else v)
    for k, v in override.items():
        setattr(cfg, k, v)
    return cfg
