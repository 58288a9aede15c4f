// lt_device.h -- device-resident data model of the engine (gfx950).
//
// HBM layout (see DESIGN.md "Data layout"):
//   particles   par(numpar,13) exactly as the reference holds it: attribute-major
//               SoA, so lane n reads par[k*numpar+n] coalesced (LTRANS.f90:104-120)
//   rho nodes   one 32-byte record {x, y, depth, angle} per node (2 x dwordx4)
//   u / v nodes one 16-byte record {x, y}
//   elements    int4 node ids per wet element; adjacency rows [element][10]
//   slabs       [node][level][slot] (slot = the reference's FIRST index of
//               t_Uvel(3,node,level), hydrodynamic_module.f90:682-689): the 4
//               levels x 3 time slots a spline window needs are 48 contiguous
//               bytes in fp32 storage instead of 12 scattered 8-byte words
//   zeta        [node][4] (3 slots + pad) -> one aligned dwordx4
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ltd {

struct RhoNode { double x, y, depth, angle; };
struct XYNode { double x, y; };

enum Counter : int {
  CNT_ACTIVE = 0,     // particles tracked this step
  CNT_ERRORS = 1,     // particles that raised an err_code
  CNT_SIGERR = 2,     // spline fits that fell back to linint (SigErr /= 0)
  CNT_BRUTE = 3,      // boundary queries that used the full segment scan
  CNT_PIP_SLOW = 4,   // point-in-polygon queries that used the full algorithm
  CNT_VT_QUEUED = 5,  // VTurb spline intervals whose tension factor needed the batched SIGS iteration
  CNT_VT_ROUNDS = 6,  // rounds (of up to 64 intervals) of that iteration, summed over the waves
  CNT_SLOW = 7,       // particle-steps k_rk listed for k_rk_slow (the list's length)
  CNT_COUNT = 8,
  // (not a per-step counter -- k_pre does not clear it: steps whose draw total differed from the one the host
  // assumed under a static stream window, lt_set_rng_window_static)
  CNT_STATIC_MISS = 8,
  CNT_ALLOC = 9
};

// uniform-grid index over boundary segments: cell (cx,cy) lists, ascending and
// de-duplicated, every segment whose bounding box overlaps the 2x2 block of
// cells (cx..cx+1, cy..cy+1).  A particle move whose bounding box spans at
// most 2 cells per axis therefore reads ONE list in reference order.
struct SegIndex {
  const int* ptr;   // (ncx*ncy+1)
  const int* ids;   // 1-based segment ids
  double x0, y0, inv_cw, inv_ch;
  int ncx, ncy;
};

// Slab index over polygon edges for the crossings test (point_in_polygon_module.f90:25-167 casts a ray towards +x and
// counts the edges that STRADDLE the point's y).  The slabs are the intervals between consecutive distinct vertex
// ordinates ys[0] < ys[1] < ...: inside one, the edges that straddle y are exactly those that span the whole slab, so
// slab k lists them (in polygon order) and nothing else -- two wall edges in the interior of a basin, one or two of a
// coastline however many of its edges lie within a few metres of the point's y.  (Uniform y-bins listed every edge whose
// y-range touched the bin: a coast running along x put hundreds of edges into the bins next to it, and every particle
// near that coast scanned them all, every step -- profiles/r05_config5_shard.)  A point whose y EQUALS a vertex ordinate
// goes to the reference's full algorithm (`slow`).  Lookup: a uniform grid of ncell cells over [ys[0], ys[m-1]] gives the
// range of ys that can hold the answer (gidx[c] = number of ys whose cell is below c; the cell function is monotone and
// evaluated identically on the host), a bisection over that range the slab.
struct PipIndex {
  const double* ys;    // (m) distinct vertex ordinates, ascending
  const int* gidx;     // (ncell+1)
  const int* ptr;      // (m) slab k -> its entries [ptr[k], ptr[k+1]); slab k = (ys[k], ys[k+1]), k < m-1
  const double4* rec;  // per entry: the edge {x1, y1, x2, y2}
  const int* pid;      // per entry: id of the edge's polygon (islands: hid), or null
  double y0, inv_h;
  int ncell, m;
};

// Physical time slots of a slab: the three records in use (back / centre / forward) and a
// spare one that the next record is prefetched into while steps run (lt_slab_prefetch).
#define LT_SLOTS 4
// Two slab layouts.  The velocity fields U, V, W, which k_rk gathers 4-level windows of, are
// stored [node][slot][level] ("level-fastest", LT_LAYOUT_LEVFAST): the window of one time slot
// is 16 contiguous bytes (fp32 storage) = ONE lane access of the vector L1, where the
// [node][level][slot] layout of the other fields (Kh, salinity, temperature: read a whole
// level of all slots at a time) needs four.  k_rk is bound by the number of lane accesses
// the L1 can take per cycle, not by bytes.
#define LT_LAYOUT_SLOTFAST 0
#define LT_LAYOUT_LEVFAST 1

// one knot of VTurb's spline table (lt_vturb.h): abscissa, ordinate, derivative.  The tension factor of the interval
// that starts at the knot is zero for nine intervals in ten and lives elsewhere: a bit per knot and particle
// (StepParams::vt_bits) says whether the interval was listed for the SIGS iteration, the factors of those are in
// StepParams::vt_sig (same indexing as the records, touched only where a bit is set).
struct VtRec { double x, y, yp; };

// per-wave task lists of VTurb's SIGS iterations (lt_vturb.h)
struct VtQueue {
  double *S, *S1, *S2;      // [wave][cap]
  unsigned short* wl;       // knot << 6 | lane
  int* count;               // [wave]
  int cap;
};

struct StepParams {
  // ---- particle state (device copies of the reference's arrays)
  double* par;
  int numpar;
  int *P_r_element, *P_u_element, *P_v_element;
  int *P_behave, *dead, *oob, *settle, *endpoly, *hitBottom, *hitLand, *err_code;
  double *P_swim, *P_pediage, *P_deadage, *settletime;
  // ---- grid statics
  const RhoNode* rnode;
  const XYNode* unode;
  const XYNode* vnode;
  const int4* RE;
  const int4* UE;
  const int4* VE;
  const int* rAdj;  // [element][10]
  const int* uAdj;
  const int* vAdj;
  const double* lev;  // 6 tables of 64 doubles: hc*SC, CS, aux (rho) ; hc*SCW, CSW, aux (w)
  int us, ws;
  int Vtransform;
  double hc, z0;
  // ---- hydrodynamic slabs
  const void *U, *V, *W, *Kh, *zeta;
  int tb, tc, tf;  // 0-based PHYSICAL slot (0..LT_SLOTS-1) of the back / centre / forward record
  // ---- time
  double ex0, ex1, ex2, ix0, ix1, ix2;
  int p, it, idt;
  // reciprocals (as double-doubles) of the divisors shared by all particles (div_by in lt_math.h):
  // ex(2)-ex(3), ex(1)-ex(2), ex(1)-ex(3), 4*ws, 6; fastdiv = 0 falls back to '/'
  double c_d23[2], c_d12[2], c_d13[2], c_p2[2], c_6[2];
  int fastdiv;
  int pns0, pns1, pns2;  // polintd's closest time node (1..3) for ix(1), ix(2), ix(3)
  // ---- options
  int Behavior, OpenOceanBoundary, mortality, settlementon, holesExist, ErrorFlag, TrackCollisions;
  double swimstart, swimfast, sink;
  // behaviours 1-5, 7 and SaltTempOn (behavior_module.f90:181-551, LTRANS.f90:1062-1076)
  int dt, SaltTempOn;
  // free slip (hydrodynamic_module.f90:1934-1994, 2329-2520): per element, bit k = node k+1 is water;
  // fsVU: per U element, v_mask at the U element's node ids (the reference's v-grid diagonal cases)
  int FreeSlip;
  const unsigned char *fsR, *fsU, *fsV, *fsVU;
  double Sgradient, Hswimspeed, Swimdepth, twistart, twiend, daylength, Em, Kd, thresh;
  const void *S, *T;                 // salinity / temperature slabs [rho node][us][LT_SLOTS]
  double *timer, *P_Sprev, *P_zprev, *P_Salt, *P_Temp;
  int* bottom;
  // ---- boundaries
  const double* bnd;   // [nbounds][4] = x1, y1, x2, y2
  const int* land;     // [nbounds]
  int nbounds;
  SegIndex seg;
  int use_seg_index;
  const double* bx; const double* by; int maxbound;
  const double* hx; const double* hy; const int* hid; int maxisland; int numislands;
  PipIndex pip_main, pip_isl;
  int use_pip_index;
  // ---- habitat
  const double* polys; const double* holes; int pedges, hedges;
  const int* polyspecs; const int* holespecs; int npoly, nhole;
  const double* maxbdis; const double* maxhdis;
  const int* elepolys_ptr; const int* elepolys_ids;
  const int* polyholes_ptr; const int* polyholes_ids;
  int minpolyid, minholeid;
  // ---- turbulence (HTurb / VTurb) and the materialised MT19937 stream
  int HTurbOn, VTurbOn;
  double ConstantHTurb, PI;
  const uint32_t* rng_ring;       // raw stream words, ring of rng_mask+1 words
  unsigned long long rng_mask;    // ring size - 1 (power of two)
  unsigned long long rng_base;    // stream index of the first draw of THIS shard's first active particle
  int rng_draws;                  // turbulence draws per active particle (4 HTurb + 2*(idt/2) VTurb);
                                  // the behaviour draws (0..3, data dependent) follow them
  const int* rank;                // [numpar] offset of the particle's first draw behind rng_base: exclusive
                                  // sum, in reference order, of the draws of the tracked particles
  const int* orig;                // [numpar] reference index of internal slot i
  long long expect_own;           // >= 0: the shard's draw total the host assumed (static window); k_post checks rank[numpar]
  double* vt_dev;   // the walk's normal deviates [group of 64 particles][idt/2 sub-steps][lane] (k_vt_norm)
  VtRec* vt_tab;    // VTurb spline table [wave][4*ws][lane] of {x, y, yp}
  double* vt_sig;   // [wave][4*ws][lane] tension factor of the interval that starts at the knot -- listed intervals only
  unsigned long long* vt_bits;   // [wave][4][lane] bit k of the lane's 256: the interval that starts at knot k is listed
  VtQueue vt_q;     // per-wave task lists of the SIGS iterations (lt_vturb.h)
  // the walks that met an interval with a tension factor and are finished by k_vt_walk (lt_vturb.h):
  // their number, the particle slots, the sub-step each resumes at
  int *vt_wl_count, *vt_wl_list, *vt_wl_resume;
  // ---- per-step workspace exchanged between k_pre / k_rk / k_post
  unsigned char* ele_ok;  // [numpar] 1: the stored elements were found for the CURRENT position
                          //     by the setEle at the end of the previous step (LTRANS.f90:1317)
  double* ws_d;   // [WS_NDOUBLE][numpar]
  int* ws_i;      // [numpar] flags | err << 8
  // ---- counters
  unsigned long long* counters;
  unsigned int* block_active;  // [blocks] tracked particles per k_pre block
  int* slow_list;              // [numpar] the particle slots k_rk hands to k_rk_slow (counters[CNT_SLOW] of them)
};

}  // namespace ltd
