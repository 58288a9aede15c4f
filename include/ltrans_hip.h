/*
 * ltrans_hip.h -- C ABI of the MI355X-native LTRANS particle-advection engine.
 *
 * This is the drop-in boundary for ONE path of LTRANS v.2b: the per-particle,
 * per-internal-timestep loop
 *
 *     subroutine update_particles()            Model/LTRANS.f90:707-1419
 *
 * and everything it calls (find_currents :1422-1614, HYDRO_MOD search/interp
 * hydrodynamic_module.f90:1414-2777, TENSION_MOD, INT_MOD, GRIDCELL_MOD,
 * HTURB_MOD, VTURB_MOD, NORM_MOD, RANDOM_MOD, BOUNDARY_MOD runtime part
 * boundary_module.f90:1515-1902, PIP_MOD, BEHAVIOR_MOD runtime part,
 * SETTLEMENT_MOD runtime part settlement_module.f90:485-640).
 *
 * The reference has no FFI: update_particles() is an internal procedure of
 * PROGRAM main with no arguments, reading host-associated variables of main
 * (par, ex, ix, p, it, hitBottom, hitLand, ... LTRANS.f90:120-126) and
 * PRIVATE SAVE module state.  The entry points below are therefore exactly
 * what a Fortran ISO_C_BINDING interface for that path binds: one call per
 * group of module state the path reads (uploaded once, at the place the
 * reference initialises it) and one call, lt_step(), at the place the
 * reference calls update_particles() (LTRANS.f90:596).
 *
 * Conventions (all taken from the reference so the Fortran host passes its
 * arrays unchanged):
 *   - INTEGER = int32_t, DOUBLE PRECISION = double, REAL = float,
 *     LOGICAL = int32_t (0 = .FALSE., non-zero = .TRUE.).
 *   - arrays are column-major with the extents written in the comments;
 *     node / element / segment ids are 1-based, 0 = empty adjacency slot.
 *   - every function returns an int status: LT_OK (0) or a negative LT_ERR_*.
 *     No host pointer is retained after a call returns.
 *   - an engine is single-caller (like the reference: one thread); it owns one
 *     HIP stream on one device.
 *   - plain pointers and sizes only: no torch / C++ types cross this ABI.
 */
#ifndef LTRANS_HIP_H
#define LTRANS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_ABI_VERSION 2

/* status codes */
#define LT_OK                 0
#define LT_ERR_ARG           -1  /* NULL / out-of-range argument                   */
#define LT_ERR_ABI           -2  /* struct_size / abi_version mismatch             */
#define LT_ERR_STATE         -3  /* call order: grid / slabs / particles missing   */
#define LT_ERR_UNSUPPORTED   -4  /* namelist value outside the reference's own range
                                    (Behavior not in 0..7)                          */
#define LT_ERR_DEVICE        -5  /* HIP runtime error (see lt_last_error)          */
#define LT_ERR_SLAB_NOT_F32  -6  /* slab_storage = F32 but a value is not a float  */
#define LT_ERR_NOMEM         -7

/* per-particle error codes written to err_code[] by lt_step().
 * 4,5,6 : particle not in any neighbouring rho/u/v element at the START of
 *         the step  (hydrodynamic_module.f90:1462-1510; ErrorLog formats 21-23,
 *         LTRANS.f90:761-763,834-879)
 * 24    : still crossing a boundary after the 3rd reflection (LTRANS.f90:1203)
 * 25    : outside the main boundary after intersect_reflect  (LTRANS.f90:1241)
 * 26    : inside an island after intersect_reflect           (LTRANS.f90:1276)
 * 27-29 : jumped over a rho/u/v element (search after the move, :1319-1364)  */
#define LT_PERR_NONE          0
#define LT_PERR_NOT_IN_RHO    4
#define LT_PERR_NOT_IN_U      5
#define LT_PERR_NOT_IN_V      6
#define LT_PERR_3RD_REFLECT  24
#define LT_PERR_OUT_MAIN     25
#define LT_PERR_IN_ISLAND    26
#define LT_PERR_JUMP_RHO     27
#define LT_PERR_JUMP_U       28
#define LT_PERR_JUMP_V       29

/* slab storage on the device */
#define LT_SLAB_F32 0  /* ROMS history values are floats: stored as fp32, checked lossless */
#define LT_SLAB_F64 1  /* stored as the reference holds them (fp64)                        */

/* Scalars of LTRANS.data that the path reads (schema LTRANS.h:32-269, read by
 * getParams parameter_module.f90:21-136).  Field names are the namelist names. */
typedef struct lt_config {
  int32_t struct_size;        /* = sizeof(lt_config)                                   */
  int32_t abi_version;        /* = LT_ABI_VERSION                                      */
  /* $numparticles */
  int32_t numpar;             /* particles held by THIS engine (the local shard)       */
  /* $timeparam */
  int32_t dt;                 /* external time step (s)                                */
  int32_t idt;                /* internal time step (s)                                */
  /* $hydroparam */
  int32_t us, ws;             /* rho / w s-levels; the path assumes ws = us+1
                                 (LTRANS.f90:949-951)                                  */
  float   hc;                 /* REAL in LTRANS.h:66                                   */
  int32_t Vtransform;         /* 1,2,3 (hydrodynamic_module.f90:2691-2777)             */
  double  z0;
  /* $turbparam */
  int32_t HTurbOn, VTurbOn;   /* hor_turb_module.f90:29 / ver_turb_module.f90:30; the single
                                 serial MT19937 stream is reproduced on the device        */
  double  ConstantHTurb;
  /* $behavparam */
  int32_t Behavior;           /* 0 passive, 1 near-surface, 2 near-bottom, 3 diurnal vertical
                                 migration, 4 C. virginica, 5 C. ariakensis, 6 constant
                                 sink/float, 7 tidal stream transport
                                 (behavior_module.f90:181-551)                            */
  int32_t OpenOceanBoundary;
  int32_t mortality;
  double  deadage, pediage, swimstart, swimslow, swimfast, sink, Swimdepth;
  double  Sgradient, Hswimspeed;                       /* rest of $behavparam          */
  /* $dvmparam (behaviour 3) */
  double  twistart, twiend, daylength, Em, Kd, thresh;
  /* $settleparam */
  int32_t settlementon, holesExist;
  /* $convparam */
  double  PI;                 /* namelist value 3.14159265358979 (norm_module.f90:37)  */
  double  Earth_Radius;       /* x2lon / y2lat of the output path                      */
  int32_t SphericalProjection;
  double  lonmin, latmin;
  /* $other */
  int32_t seed;               /* init_genrand(seed) (LTRANS.f90:241)                   */
  int32_t ErrorFlag;          /* 0 stop, 1 return to previous, 2 kill, 3 set out       */
  int32_t SaltTempOn;         /* P_Salt / P_Temp at the particle (LTRANS.f90:1062-1076) */
  int32_t TrackCollisions;
  int32_t FreeSlip;           /* free-slip substitution at land nodes (hydrodynamic_module.f90:1934-1994,
                                 2329-2520); needs the masks in lt_grid                   */
  /* build-specific (not in LTRANS.data) */
  int32_t slab_storage;       /* LT_SLAB_F32 / LT_SLAB_F64                             */
  int32_t device;             /* HIP device ordinal, -1 = current device               */
  int32_t resort_every;       /* internal steps between re-sorts of the device-side particle order by
                                 rho element (keeps the gathers of a wave local as particles mix);
                                 0 = default (32), < 0 = never.  No effect on results.     */
} lt_config;

/* Grid statics produced by initGrid (hydrodynamic_module.f90:88-657). */
typedef struct lt_grid {
  int32_t rho_nodes, u_nodes, v_nodes;
  int32_t rho_elements, u_elements, v_elements;
  const double  *rx, *ry;        /* (rho_nodes)                  :58,124-125        */
  const double  *ux, *uy;        /* (u_nodes)                                       */
  const double  *vx, *vy;        /* (v_nodes)                                       */
  const double  *depth;          /* (rho_nodes) positive down    :38,116            */
  const double  *rho_angle;      /* (rho_nodes)                  :58,123            */
  const int32_t *rho_mask, *u_mask, *v_mask; /* FreeSlip and lt_slab_prefetch; else may be NULL */
  const int32_t *RE, *UE, *VE;   /* (4,elements) node ids        :49,117-119        */
  const int32_t *r_Adjacent;     /* (rho_elements,10) self first, 0-terminated :54  */
  const int32_t *u_Adjacent;     /* (u_elements,10)                                 */
  const int32_t *v_Adjacent;     /* (v_elements,10)                                 */
  const double  *SC, *CS;        /* (us)                         :35,112-113        */
  const double  *SCW, *CSW;      /* (ws)                         :35,114-115        */
} lt_grid;

/* Boundary data produced by createBounds (boundary_module.f90:1148-1204). */
typedef struct lt_bounds {
  int32_t nbounds;               /* boundary segments                               */
  int32_t maxbound;              /* points of the closed main polygon               */
  int32_t maxisland;             /* points of all closed island polygons            */
  int32_t numislands;
  const double  *bnd_x, *bnd_y;  /* (2,nbounds)                  :41                */
  const int32_t *land;           /* (nbounds) LOGICAL            :44                */
  const double  *bx, *by;        /* (maxbound)                   :39                */
  const double  *hx, *hy;        /* (maxisland)                                     */
  const int32_t *hid;            /* (maxisland) island id per point :40             */
} lt_bounds;

/* Habitat polygons as organised by createPolySpecs (settlement_module.f90:
 * 29-33,49-89,245-480).  Ragged Fortran derived types are passed as CSR. */
typedef struct lt_habitat {
  int32_t pedges, hedges;
  int32_t minpolyid, maxpolyid, minholeid, maxholeid;
  const double  *polys;          /* (pedges,5): id, centre x,y, edge x,y            */
  const double  *holes;          /* (hedges,6): id, centre x,y, edge x,y, poly id   */
  const int32_t *polyspecs;      /* (npoly,2) first row (1-based), #points; npoly = maxpolyid-minpolyid+1 */
  const int32_t *holespecs;      /* (nhole,2)                                       */
  const double  *maxbdis;        /* (npoly)                                         */
  const double  *maxhdis;        /* (nhole)                                         */
  const int32_t *elepolys_ptr;   /* (rho_elements+1) CSR offsets (0-based)          */
  const int32_t *elepolys_ids;   /* polygon ids listed per rho element              */
  const int32_t *polyholes_ptr;  /* (npoly+1) CSR offsets (0-based)                 */
  const int32_t *polyholes_ids;  /* hole ids listed per polygon                     */
} lt_habitat;

/* Per-particle state.  Every pointer may be NULL = "do not transfer". */
typedef struct lt_particles {
  double  *par;                  /* par(numpar,13)               LTRANS.f90:104-120 */
  int32_t *P_r_element;          /* (numpar) hydrodynamic_module.f90:23             */
  int32_t *P_u_element;
  int32_t *P_v_element;
  int32_t *P_behave;             /* behavior_module.f90:64                          */
  int32_t *dead;                 /* LOGICAL :83                                     */
  int32_t *oob;                  /* LOGICAL :86                                     */
  double  *P_swim;               /* (numpar,3) :77                                  */
  double  *P_pediage;            /* :68                                             */
  double  *P_deadage;            /* :69                                             */
  int32_t *settle;               /* LOGICAL settlement_module.f90:32                */
  double  *settletime;           /* :31                                             */
  int32_t *endpoly;              /* LTRANS.f90:122                                  */
  int32_t *hitBottom;            /* LTRANS.f90:122 (TrackCollisions)                */
  int32_t *hitLand;
  int32_t *err_code;             /* LT_PERR_* of the last step (build addition)     */
  /* behaviours 4, 5, 7 (behavior_module.f90:61,70-71,80) */
  double  *timer;                /* C. ariakensis downward-swimming timer           */
  double  *P_Sprev;              /* salinity at the previous location               */
  double  *P_zprev;              /* previous depth                                  */
  int32_t *bottom;               /* LOGICAL: on the bottom (behaviour 7)            */
  /* SaltTempOn (LTRANS.f90:124,1072-1075): outputs only */
  double  *P_Salt;
  double  *P_Temp;
} lt_particles;

/* One time record of the hydrodynamic fields, as updateHydro scatters it into
 * slot t_f (hydrodynamic_module.f90:1371-1403): values already multiplied by
 * the land mask, node-numbered, level-last. */
typedef struct lt_slab {
  const double *zeta;            /* (rho_nodes)                                     */
  const double *Uvel;            /* (u_nodes,us)                                    */
  const double *Vvel;            /* (v_nodes,us)                                    */
  const double *Wvel;            /* (rho_nodes,ws)                                  */
  const double *Kh;              /* (rho_nodes,ws)  NULL allowed when VTurbOn = 0   */
  const double *salt;            /* (rho_nodes,us)  needed by behaviours 4,5,7 and SaltTempOn */
  const double *temp;            /* (rho_nodes,us)  needed by SaltTempOn            */
} lt_slab;

/* One ROMS history record as updateHydro holds it right after NF90_GET_VAR
 * (hydrodynamic_module.f90:1065-1368), BEFORE the mask multiply and the copy into
 * t_*(t_f,node,level) of :1371-1403, which the engine does on the device:
 * romZf(vi,uj), romUf(ui,uj,us), romVf(vi,vj,us), romWf(vi,uj,ws), romKHf(vi,uj,ws),
 * Fortran order.  (i,j) -> node is count = (j-1)*vi + i (rho, v) and (j-1)*ui + i (u),
 * so the arrays are node-numbered already.  dtype selects how the host read them:
 * LTRANS allocates DOUBLE PRECISION arrays; a host that lets NetCDF deliver REAL
 * arrays (the file's own type) halves the PCIe bytes.  window = t_ijruv(12)
 * (hydrodynamic_module.f90:42, setijruv :2851-2922): only
 *   rho / w / Kh / zeta  i = w[0]..w[1], j = w[2]..w[3]
 *   u                    i = w[4]..w[5], j = w[6]..w[7]
 *   v                    i = w[8]..w[9], j = w[10]..w[11]      (1-based, inclusive)
 * are valid in the host arrays and are refreshed on the device; the other nodes of
 * the slot keep what they held, as the reference's t_* arrays do. */
#define LT_ROM_F64 0
#define LT_ROM_F32 1
typedef struct lt_rom_record {
  int32_t struct_size;           /* = sizeof(lt_rom_record)                          */
  int32_t dtype;                 /* LT_ROM_F64 | LT_ROM_F32 : type of the five arrays */
  int32_t vi, uj, ui, vj;        /* grid dimensions (PARAM_MOD)                      */
  int32_t masked;                /* 0: raw ROMS values, the engine multiplies by the
                                  *    node masks of lt_grid (m_r, m_u, m_v);
                                  * 1: already multiplied                            */
  int32_t has_window;            /* 0: whole grid; 1: window[] applies               */
  int32_t window[12];            /* t_ijruv                                          */
  const void *romZ, *romU, *romV, *romW;
  const void *romKH;             /* NULL allowed when VTurbOn = 0                    */
  const void *romS, *romT;       /* (vi,uj,us) salinity / temperature, as lt_slab     */
} lt_rom_record;

/* What printOutput / writeOutput hand to the CSV and NetCDF writers (LTRANS.f90:1617-1779):
 * longitude / latitude of the particle (x2lon, y2lat, conversion_module.f90:322-378), depth,
 * age, getStatus (behavior_module.f90:554-574: behaviour id, -1 dead, -2 settled, -3 out of
 * bounds; a DOUBLE PRECISION array there), and the optional columns.  Host arrays of numpar
 * values in the host's particle order; NULL = not wanted. */
typedef struct lt_output {
  double  *lon, *lat, *z, *age, *status;
  double  *salt, *temp;          /* SaltTempOn                                       */
  int32_t *hitBottom, *hitLand;  /* TrackCollisions                                  */
} lt_output;

/* Per-step counters (sums over the shard). */
typedef struct lt_step_stats {
  int64_t active;                /* particles that were tracked this step           */
  int64_t errors;                /* particles with err_code != 0                    */
  int64_t rng_draws;             /* MT19937 draws consumed by this shard            */
} lt_step_stats;

typedef struct lt_engine lt_engine;

/* lifecycle ------------------------------------------------------------- */
int lt_create(const lt_config *cfg, lt_engine **out);
int lt_destroy(lt_engine *e);
const char *lt_last_error(const lt_engine *e);     /* NUL-terminated, engine-owned */
int lt_abi_version(void);
/* hash of the kernel sources the library was built from (profiles/<tag>/pmc.json records the one
 * its counters were collected with; "unknown" for a build outside __graft_entry__.build()) */
const char *lt_source_hash(void);

/* uploads: replace the module state the path reads ----------------------- */
int lt_set_grid(lt_engine *e, const lt_grid *g);            /* after initGrid      */
int lt_set_bounds(lt_engine *e, const lt_bounds *b);        /* after createBounds  */
int lt_set_habitat(lt_engine *e, const lt_habitat *h);      /* after initSettlement*/
int lt_set_particles(lt_engine *e, const lt_particles *p);  /* after setEle_all    */
/* slot = 1,2,3 = first index of t_zeta(3,:) etc.  initHydro loads slots 1,2,3;
 * updateHydro rewrites slot t_f only. */
int lt_set_slab(lt_engine *e, int32_t slot, const lt_slab *s);
/* rotating indices t_b,t_c,t_f (hydrodynamic_module.f90:42,741-743,1080-1082) */
int lt_set_time_slots(lt_engine *e, int32_t t_b, int32_t t_c, int32_t t_f);

/* The slab pipeline (the rotation + upload of updateHydro, hydrodynamic_module.f90:
 * 1052-1410, as an asynchronous double buffer).  The device keeps FOUR physical time
 * slots per field: the three in use and a spare one.
 *   lt_slab_prefetch  copies the record into pinned staging (the only part the caller
 *                     waits for: its arrays may be reused on return), then H2D copy,
 *                     mask multiply, type conversion and scatter into the SPARE slot run
 *                     on a copy stream beside the steps of the current external step.
 *   lt_slab_commit    = the rotation: the prefetched record becomes slot t_f, the slot
 *                     that held the old back record becomes the spare, the time roles
 *                     are (t_b, t_c, t_f) as after hydrodynamic_module.f90:1080-1082.
 *                     Later steps are ordered after the prefetch on the device; the call
 *                     itself waits for the prefetch only to report LT_ERR_SLAB_NOT_F32.
 * lt_set_slab + lt_set_time_slots remain the synchronous equivalent. */
int lt_slab_prefetch(lt_engine *e, const lt_rom_record *r);
int lt_slab_commit(lt_engine *e, int32_t t_b, int32_t t_c, int32_t t_f);

/* Mersenne Twister state (random_module.f90:98-99): mt(0:623), mti. */
int lt_set_rng(lt_engine *e, const int32_t *mt624, int32_t mti);
int lt_get_rng(lt_engine *e, int32_t *mt624, int32_t *mti);
/* Multi-GPU: this shard's place in the single serial stream of one step:
 * number of ACTIVE particles (those that draw) held by lower-ranked shards
 * and the total over all shards.  Defaults: 0 and the shard's own count. */
int lt_count_active(lt_engine *e, const double *ix3, int64_t *n_active);
int lt_set_rng_window(lt_engine *e, int64_t active_before, int64_t active_total);
/* The same window for every following step, until a window is set again: for runs in which no
 * particle can stop drawing (nothing unreleased, no settlement, mortality or open boundary), where
 * the shards' counts never change and need not be exchanged before each step.  The shard's draws are
 * read back in the first step after this call only; the following lt_step calls enqueue the step
 * without waiting for the device, and the device compares every step's draw total with that figure:
 * if a particle did stop (or start) drawing, lt_get_stats / lt_get_counters / lt_get_particles /
 * lt_get_output return LT_ERR_STATE, lt_pack_positions / lt_allgather_positions do so one call later (they
 * do not wait for the device), and the lt_set_rng_window that ends the static window reports it once (the
 * new window is set all the same). */
int lt_set_rng_window_static(lt_engine *e, int64_t active_before, int64_t active_total);
/* Behaviours 1-5 draw a data-dependent number of deviates per particle
 * (genrand_real1, behavior_module.f90:256-452), so a shard's place in the stream is a
 * number of DRAWS: lt_count_draws runs the first part of the step (the gates, the element
 * search, the salinity cue) and returns the draws this shard will consume; the following
 * lt_step with the same arguments continues from there.  lt_set_rng_window_draws gives
 * the draws consumed by lower-ranked shards and by all shards in this step. */
int lt_count_draws(lt_engine *e, const double *ex3, const double *ix3, int32_t p, int32_t it,
                   int64_t *n_draws);
int lt_set_rng_window_draws(lt_engine *e, int64_t draws_before, int64_t draws_total);
/* The per-step exchange itself ("one tiny all-gather of per-GPU counts per step before the turbulence
 * stage when any particle can be inactive"; the skip rule is LTRANS.f90:790-816, the one serial stream
 * random_module.f90:166-197), for a host with no other communication library: runs the first part of the coming
 * step on the device as lt_count_draws does, all-gathers the shards' draw totals -- one int64 per rank,
 * ncclAllGather on the communicator of lt_comm_init, device to device -- and places this shard behind the draws of
 * the lower ranks; the following lt_step with the SAME arguments continues from there.  Works for every
 * configuration that draws (turbulence, behaviours 1-5, settlement / mortality / open boundary / late releases),
 * ends a static window, and without a communicator is the single engine's own window.  Every rank of the
 * communicator must make the call for every step.  The call itself does not wait: the G counts travel to the host
 * behind the collective and the lt_step that follows waits for them once, before it places the generators of the
 * step's random words (the one host wait a step without a static window has anyway, for the shard's own total).
 * Under lt_set_rng_window_static there is no exchange and no wait. */
int lt_exchange_rng_window(lt_engine *e, const double *ex3, const double *ix3, int32_t p, int32_t it);

/* the hot path ---------------------------------------------------------- */
/* = update_particles() with main's ex(3), ix(3), p, it (LTRANS.f90:568-571,
 * 588-590).  Enqueued on the engine's stream and returns LT_OK at once; the
 * step's counters (tracked particles, per-particle error count) are read with
 * lt_get_stats(), which synchronises. */
int lt_step(lt_engine *e, const double *ex3, const double *ix3, int32_t p, int32_t it);
int lt_sync(lt_engine *e);
int lt_get_stats(lt_engine *e, lt_step_stats *out);

/* downloads ------------------------------------------------------------- */
int lt_get_particles(lt_engine *e, lt_particles *out);
/* The conversion loop of writeOutput (LTRANS.f90:1719-1723) on the device and the download
 * of only what the writers need.  reset_hits != 0 also clears hitBottom / hitLand as
 * printOutput does after LandHits.csv / BottomHits.csv (:1665-1666). */
int lt_get_output(lt_engine *e, lt_output *out, int32_t reset_hits);
/* Pack (x,y,z) of the shard as xyz(numpar,3) column-major into DEVICE memory
 * dst (e.g. the send buffer of an RCCL all-gather); asynchronous. */
int lt_pack_positions(lt_engine *e, void *dst_device);
/* the engine's HIP stream (hipStream_t) for ordering external work */
void *lt_stream(lt_engine *e);

/* Multi-GPU: one process (one engine) per GPU, particles sharded by contiguous index block,
 * fields replicated.  The one collective of the path -- the all-gather of the shards' new (x,y,z), which
 * the reference needs as one array at print steps (LTRANS.f90:604-612) -- is done here with RCCL over xGMI,
 * so that a Fortran host needs no other library.  The RCCL library is opened at lt_comm_init.
 *   lt_comm_unique_id   (rank 0) 128 opaque bytes that every rank must pass to lt_comm_init; how they
 *                       travel is the host's business (a file, MPI_Bcast, ...)
 *   lt_comm_init_file   the same through a file: rank 0 creates `path` (atomically), the others wait
 *                       for it up to timeout_s seconds; rank 0 removes the file once the communicator
 *                       exists, and a file last written more than timeout_s before the call -- left by an
 *                       earlier or crashed run -- is ignored (so the ranks must call within timeout_s of each
 *                       other).  Use a path that is unique per run: a fixed path shared by successive runs
 *                       is NOT safe (a rank may read the id of a run that crashed seconds ago).
 *   lt_allgather_positions  counts(nranks) = numpar of every rank; result xyz(sum(counts),3), column-major,
 *                       shards in rank order, each in the reference's particle order, into host memory
 *                       (xyz_host: the call returns when the array is complete) and / or device memory
 *                       (xyz_device: asynchronous); either may be NULL.  The collective runs on a stream
 *                       of its own behind the packing of the shard: an lt_step enqueued after the call
 *                       runs beside the transfer.
 *   lt_allgather_wait   returns when the last lt_allgather_positions has delivered (xyz_device complete). */
int lt_comm_unique_id(void *id128);
int lt_comm_init(lt_engine *e, const void *id128, int32_t nranks, int32_t rank);
int lt_comm_init_file(lt_engine *e, const char *path, int32_t nranks, int32_t rank, int32_t timeout_s);
int lt_allgather_positions(lt_engine *e, const int32_t *counts, double *xyz_host, void *xyz_device);
int lt_allgather_wait(lt_engine *e);
int lt_comm_destroy(lt_engine *e);

/* init-time geometry on the device ------------------------ */
/* The adjacency lists of initGrid (hydrodynamic_module.f90:588-645): for element i the
 * elements j of [i-(row+2), i+row+2] that share a node with it, ascending, after i itself;
 * row = vi for the rho and v grids, ui for the u grid.  E is (4,elements), adjacent is
 * (elements,10) in Fortran order, zero filled.  Needs no engine; device = HIP ordinal or -1.
 * LT_ERR_ARG when an element has more than 9 such neighbours (the reference would write
 * past column 10). */
int lt_build_adjacency(int32_t device, int32_t elements, int32_t row, const int32_t *E,
                       int32_t *adjacent);
/* setEle_all (hydrodynamic_module.f90:1536-1676) for n positions given in the reference's
 * particle order, on the grid of lt_set_grid: particle 1 by a scan of all elements, particle
 * n by the adjacency row of particle n-1's element, then a scan of all elements -- the
 * element tests (gridcell) run on the device for all particles at once, the dependence on
 * the previous particle (it only matters for a point on an edge shared by elements) is
 * resolved afterwards.  *err = 0, or 1/2/3 = not in any rho/u/v element with *par = that
 * particle (1-based), as the reference's optional arguments; elements of particles from
 * *par on are 0 where the reference leaves them unset.  Any output pointer may be NULL. */
int lt_locate_particles(lt_engine *e, int32_t n, const double *Xpar, const double *Ypar,
                        int32_t *P_r_element, int32_t *P_u_element, int32_t *P_v_element,
                        int32_t *err, int32_t *par);
/* createPolySpecs (settlement_module.f90:245-480): from the edge-point tables of h (polys,
 * holes, pedges, hedges, the id ranges, maxbdis, maxhdis) builds polyspecs, holespecs, the
 * habitat polygons of every rho element (the O(rho_elements x pedges) gridcell / inpoly
 * part, on the device) and the holes of every polygon, and installs the habitat as
 * lt_set_habitat does; the list members of h are ignored and may be NULL.
 * lt_get_polyspecs copies the lists out (any pointer may be NULL; *_cap = capacity of the
 * id arrays, LT_ERR_ARG if too small -- the ptr arrays, which are always filled first,
 * then tell the size needed). */
/* createBounds (boundary_module.f90:82-1215): from mask_rho(vi,uj) and the coordinates of the u points
 * x_u,y_u(ui,uj) and v points x_v,y_v(vi,vj), the boundary arrays of lt_bounds in the order the reference's
 * clockwise walk produces them: bx,by(maxbound) closed main polygon; hx,hy,hid(maxisland) closed island
 * polygons, ids from 1001; bnd_x,bnd_y(2,nbounds) the segments; land(nbounds) = 0 on open ocean.  A serial walk
 * on the host, no engine and no device needed.  cap = capacity in points of every output array (2 x cap values
 * for bnd_x / bnd_y); vi*uj is always enough.  LT_ERR_ARG for a mask the reference cannot handle (no water,
 * more than one body of water). */
int lt_create_bounds(int32_t vi, int32_t uj, const int32_t *mask_rho, const double *x_u, const double *y_u,
                     const double *x_v, const double *y_v, int32_t cap, int32_t *nbounds, int32_t *maxbound,
                     int32_t *maxisland, int32_t *numislands, double *bx, double *by, double *hx, double *hy,
                     int32_t *hid, double *bnd_x, double *bnd_y, int32_t *land);
int lt_create_polyspecs(lt_engine *e, const lt_habitat *h);
int lt_get_polyspecs(lt_engine *e, int32_t *polyspecs, int32_t *holespecs,
                     int32_t *elepolys_ptr, int32_t *elepolys_ids, int32_t elepolys_cap,
                     int32_t *polyholes_ptr, int32_t *polyholes_ids, int32_t polyholes_cap);

/* Per-phase GPU time of the steps: the counterpart of WriteModelTiming (LTRANS.h:261; CPU_TIME stamps at
 * LTRANS.f90:780,979,1085,1096,1108,1121,1390 accumulated into timeCounts, written to Timing.csv at :1669-1689).
 * lt_phase_timing(e, 1) arms it; lt_get_phase_times returns (and with reset /= 0 clears) the accumulated
 * milliseconds of ms4 = { set-up, advection, vertical turbulence, update } -- the reference's "Set Up",
 * "Advection", "VTurb" and "Update" columns; its "HTurb" and "Behavior" work is inside the update kernel,
 * its "Hydro" column is the host's own time around lt_set_slab / lt_slab_commit. */
int lt_phase_timing(lt_engine *e, int32_t on);
int lt_get_phase_times(lt_engine *e, double *ms4, int32_t reset);

/* measurement aids (not part of the reference interface) ------------------- */
/* HIP-event timing of the step kernel on the engine's stream: arm up to
 * max_steps (start, stop) pairs, then synchronise and sum them. */
int lt_profile_begin(lt_engine *e, int32_t max_steps);
int lt_profile_end(lt_engine *e, double *total_ms, int32_t *launches);
/* diagnostics of the last step: [0] tracked particles [1] particles that raised an error [2] SigErr fallbacks
 * [3] brute-force boundary scans [4] full point-in-polygon evaluations [5] VTurb spline intervals iterated
 * [6] rounds of that iteration [7] particle-steps the RK4 kernel handed to its reference-shaped redo */
int lt_get_counters(lt_engine *e, uint64_t *out8);
/* device-side re-sorts of the particle order so far (one every resort_every steps inside lt_step) and the GPU
 * milliseconds of the last one (-1: none yet) */
int lt_get_resort_stats(lt_engine *e, int64_t *resorts, double *last_ms);
/* (host only, no device) the double-double 1/b = hl2[0] + hl2[1] the kernels multiply by instead of dividing by a
 * divisor every particle shares (4*ws, 6, differences of the external times); returns 1, or 0 when b is not of the
 * form for which that is exact for every numerator (the kernels then divide).  Exported so that the claim can be
 * checked in exact arithmetic without a GPU (tests/test_host.py). */
int lt_div_const(double b, double *hl2);

#ifdef __cplusplus
}
#endif
#endif /* LTRANS_HIP_H */
