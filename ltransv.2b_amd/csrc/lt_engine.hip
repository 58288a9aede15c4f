// lt_engine.hip -- host side of the C ABI (include/ltrans_hip.h) for gfx950.
//
// Owns the device mirrors of the module state that update_particles() reads
// (LTRANS.f90:707-1419), builds the boundary indexes once per lt_set_bounds,
// and launches the step kernel of lt_kernels.h.  There is NO CPU fallback: every
// entry point that computes runs on the HIP device or fails with LT_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is opened at lt_comm_init (no link-time dependency)
#include <unistd.h>
#include <sys/stat.h>
#include <ctime>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <map>
#include <type_traits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/ltrans_hip.h"
#include "lt_rk_launch.h"  // k_rk, k_vt_table, k_vt_norm and k_vt_walk_fast live in lt_rk_kernel.hip (its own compiler options)
#define LT_VT_DEFINE_B
#include "lt_kernels.h"
#include "lt_bounds_host.h"
#include "lt_init_host.h"
#include "lt_mt.h"

using namespace ltd;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct HostIndex {
  std::vector<int> ptr, ids;
  // point-in-polygon slab index (lt_device.h: PipIndex)
  std::vector<double> ys, rec;   // distinct vertex ordinates; {x1, y1, x2, y2} per list entry
  std::vector<int> gidx, pid;    // lookup grid; the polygon's id per list entry (islands)
};

}  // namespace

struct lt_engine {
  lt_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  // grid
  bool have_grid = false, have_bounds = false, have_habitat = false, have_particles = false;
  int rho_nodes = 0, u_nodes = 0, v_nodes = 0, rho_elements = 0, u_elements = 0, v_elements = 0;
  DevBuf rnode, unode, vnode, RE, UE, VE, rAdj, uAdj, vAdj, lev, fsR, fsU, fsV, fsVU;
  // slabs
  DevBuf U, V, W, Kh, zeta, S, T, staging, flags;
  bool slab_set[4] = {false, false, false, false};
  int t_b = 1, t_c = 2, t_f = 3;
  // bounds
  DevBuf bnd, land, bx, by, hx, hy, hid, seg_ptr, seg_ids, pm_ptr, pm_ids, pi_ptr, pi_ids, pm_rec, pi_rec, pi_pid, pm_ys, pi_ys;
  int nbounds = 0, maxbound = 0, maxisland = 0, numislands = 0;
  SegIndex seg{};
  PipIndex pip_main{}, pip_isl{};
  int use_seg_index = 1, use_pip_index = 1;
  // habitat
  DevBuf polys, holes, polyspecs, holespecs, maxbdis, maxhdis, ep_ptr, ep_ids, ph_ptr, ph_ids;
  lt_habitat hab{};
  int npoly = 0, nhole = 0;
  // host copies for the init-time geometry functions (lt_locate_particles, lt_get_polyspecs)
  std::vector<double> h_nx[3], h_ny[3];      // node coordinates of the rho / u / v grids
  std::vector<int32_t> h_E[3];               // (4,elements) node ids
  std::vector<int> h_adj[3];                 // [element][10]
  std::vector<int32_t> h_polyspecs, h_holespecs, h_ep_ptr, h_ep_ids, h_ph_ptr, h_ph_ids;
  // particles
  DevBuf ws_d, ws_i, orig, ele_ok, block_active, count_slots, out_cols, out_hits, get_stage;
  size_t get_stage_used = 0;
  // pinned host staging of the downloads (get_fields_*): the device-to-host copies land here at the link's rate and
  // a few host threads move each array on to the caller's (pageable) memory while the next one arrives
  void* get_pin = nullptr;
  size_t get_pin_bytes = 0;
  bool get_use_pin = false;   // this download goes through get_pin (decided by get_fields_begin)
  struct GetItem { const void* src; void* dst; size_t bytes; };
  std::vector<GetItem> get_items;
  std::vector<hipEvent_t> get_ev;
  // internal particle order: slot i of every device array holds the reference's
  // particle perm[i] (0-based); sorted by rho element for memory locality
  std::vector<int> perm;
  bool sort_particles = true;
  // periodic re-sort on the device (resort()): the host copy of perm is refreshed lazily
  bool perm_stale = false;
  int resort_every = 32, steps_since_sort = 0;
  DevBuf sort_keys, sort_keys_in, sort_idx_in, sort_idx_out, sort_tmp, permute_tmp, tile_key;
  std::vector<int32_t> h_tile_key;   // sort key of every rho element (tiles of 32 x 32 elements), host copy
  int tile_key_bits = 1;
  DevBuf par, pr, pu, pv, P_behave, dead, oob, settle, endpoly, hitBottom, hitLand, err_code,
      P_swim, P_pediage, P_deadage, settletime, timer, P_Sprev, P_zprev, bottom, P_Salt, P_Temp;
  // rng: the reference's single serial MT19937 stream, materialised on the device
  // in blocks (lt_mt.h).  Stream word k ("virtual stream"): k < 624 -> mt[k] of the
  // state given at start, k >= 624 -> generated; the cursor starts at mti.
  uint32_t mt[624];
  int32_t mti = 624;
  int64_t rng_before = 0, rng_total = -1;
  bool rng_kernels_configured = false;
  // per-phase GPU time (the counterpart of WriteModelTiming's CPU_TIME stamps, LTRANS.f90:780-1397)
  bool phase_on = false;
  std::vector<hipEvent_t> phase_ev;   // 5 events per recorded step: before k_pre | after set-up | after RK4 | after VTurb | after k_post
  size_t phase_used = 0;
  double phase_ms[4] = {0.0, 0.0, 0.0, 0.0};   // set-up, advection, vertical turbulence, update
  // multi-GPU: RCCL communicator of lt_comm_init (nullptr = none)
  ncclComm_t comm = nullptr;
  int comm_nranks = 0, comm_rank = 0;
  DevBuf comm_send, comm_recv, comm_out, comm_counts;
  hipStream_t comm_stream = nullptr;   // the collective runs beside the next step
  hipEvent_t comm_packed = nullptr, comm_done = nullptr;
  bool comm_busy = false;              // comm_done has been recorded
  // lt_exchange_rng_window: the shards' draw counts of the coming step, all-gathered on the communicator's stream
  DevBuf cnt_dev;                      // [0] this shard's draws (int64), [8 ..] the G gathered counts
  long long* h_cnt = nullptr;          // pinned: the gathered counts as the host reads them (one read for all ranks)
  int h_cnt_cap = 0;
  hipEvent_t cnt_ready = nullptr;      // the prefix sum of the coming step has run (engine's stream)
  bool pre_defer_read = false;         // step_prepare leaves the shard's draw total on the device (the exchange reads it)
  // The host needs a step's draw totals to place the generators of the step's random words.  They travel to the host
  // asynchronously (the shard's own: own_ev behind the prefix sum; the job's: cnt_arrived behind the all-gather) and
  // whoever needs them first waits for them ONCE (resolve_own): lt_step before it enqueues the generators, lt_count_draws
  // at once.
  int* h_own = nullptr;                // pinned: the shard's draw total (rank[n]) of the coming step
  hipEvent_t own_ev = nullptr;
  bool own_pending = false;            // the copy into h_own has been enqueued, nobody has waited for it
  hipEvent_t cnt_arrived = nullptr;    // comm_stream: the gathered counts are in h_cnt
  bool xchg_pending = false;
  int xchg_G = 1, xchg_rank = 0;
  bool rng_static = false;  // lt_set_rng_window_static: the window holds until it is set again
  int64_t rng_static_before = 0, rng_static_total = -1;
  int64_t rng_static_own = -1;     // the shard's draws per step under the static window, read back ONCE (the first
                                   // step after the window was set); the device checks every later step against it
  hipEvent_t step_done = nullptr;  // end of the last lt_step on `stream`: the generators of the next step wait for it
  std::vector<hipEvent_t> rng_job_ev;  // per job slot: the launch that read it has finished
  bool rng_ready = false;
  uint64_t rng_cur = 624;          // next stream word to be consumed
  int rng_P = 256;                 // generator states
  int rng_R = 0;                   // ring capacity in blocks (power of two)
  DevBuf rng_states, rng_ring, rng_polys;
  std::vector<long long> rng_pos;  // the block each generator state is positioned at
  bool rng_gen0_raw = true;        // state 0 is still the given mt array (word 624, not a block start)
  std::map<long long, int> rng_poly_slot;  // jump distance in blocks -> slot of x^(d*S) mod phi in rng_polys
  std::vector<uint32_t> rng_g0;    // x^(S-624) mod phi: the mt array to the start of block 1
  std::vector<uint32_t> rng_gS, rng_gPm1;            // x^S, x^((P-1)*S)
  std::vector<std::vector<uint32_t>> rng_pow_PS;     // x^(j*P*S), j = 1, 2, ...
  bool rng_family_done = false, rng_rebased = false, rng_chain_done = false;
  ltmt::BlockJob* rng_jobs = nullptr;  // pinned, mapped: RNG_JOB_SLOTS launches of up to P jobs
  int rng_job_slot = 0;
  std::vector<long long> rng_ring_blk;  // the stream block each ring slot holds (-1: none)
  DevBuf flag, rank, scan_tmp, vt_tab, vt_bits, vt_qS, vt_qS1, vt_qS2, vt_qwl, vt_qcount, vt_wl, vt_dev, slow_list;
  int vt_qcap = 0;
  size_t scan_tmp_bytes = 0;
  // counters
  DevBuf counters;
  unsigned long long* h_counters = nullptr;  // pinned
  lt_step_stats stats{};
  bool stats_valid = true, counters_fetched = true;
  // the static-window word read back without a wait (check_static_miss_async): pinned copy, its event, in flight?
  unsigned long long* h_miss = nullptr;
  hipEvent_t miss_ev = nullptr;
  bool miss_inflight = false;
  bool miss_reported = false;      // a broken static window has been reported to the caller (by any call)
  bool quiet_window_switch = false;
  // optional HIP-event timing of the step kernel (lt_profile_*)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;  // pairs (start, stop)
  // slab pipeline (lt_slab_prefetch / lt_slab_commit): logical slot 1..3 -> physical slot,
  // the spare physical slot, a copy stream, pinned staging (double buffer)
  int phys[4] = {0, 0, 1, 2};
  int spare = 3;
  hipStream_t copy_stream = nullptr;
  hipEvent_t prefetch_done = nullptr, compute_mark = nullptr, pin_free[2] = {nullptr, nullptr};
  bool prefetch_pending = false, mark_valid = false, pin_busy[2] = {false, false};
  void* pin[2] = {nullptr, nullptr};
  size_t pin_bytes = 0;
  int pin_next = 0;
  DevBuf stage_rom, rmask, umask, vmask;
  bool have_masks = false;
  // first part of a step done ahead by lt_count_draws (or by lt_step itself)
  StepParams pre_P;
  bool pre_valid = false;
  int64_t pre_own = 0;
  bool pre_synced = true;   // step_prepare waited for the device (everything earlier on `stream` has completed)
  double pre_sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // VTurb's kernels behind the table (launch_vturb_head / _tail): k_vt_sigs on a stream of its own beside k_vt_norm and
  // k_vt_walk_fast (vt_aside; LT_VT_ASIDE=0 puts it back into the chain, for measurements)
  hipStream_t vt_stream = nullptr;
  hipEvent_t vt_ev[2] = {nullptr, nullptr};
  int vt_aside = 1;
  hipStream_t rng_stream = nullptr;  // the MT19937 blocks of a step are generated beside k_rk
  hipEvent_t rng_done = nullptr;
  bool rng_pending = false, step_done_valid = false;
  hipEvent_t resort_ev[2] = {nullptr, nullptr};  // around the last device-side re-sort (lt_get_counters reports its time)
  unsigned resorts = 0;
  size_t prof_used = 0;
};

namespace {

int fail(lt_engine* e, int code, const char* fmt, ...) {
  if (e) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e->err, sizeof(e->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HIPCHK(e, call)                                                                    \
  do {                                                                                     \
    hipError_t _r = (call);                                                                \
    if (_r != hipSuccess)                                                                  \
      return fail(e, LT_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), \
                  __FILE__, __LINE__);                                                     \
  } while (0)

int dev_alloc(lt_engine* e, DevBuf& b, size_t bytes) {
  if (b.p && b.bytes >= bytes && bytes > 0) return LT_OK;
  if (b.p) {
    (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
  }
  size_t n = bytes ? bytes : 16;
  HIPCHK(e, hipMalloc(&b.p, n));
  b.bytes = n;
  return LT_OK;
}

int upload(lt_engine* e, DevBuf& b, const void* src, size_t bytes) {
  int rc = dev_alloc(e, b, bytes);
  if (rc) return rc;
  if (bytes) HIPCHK(e, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, e->stream));
  return LT_OK;
}

#define TRY(x)            \
  do {                    \
    int _rc = (x);        \
    if (_rc) return _rc;  \
  } while (0)

// adjacency: (elements,10) column-major -> [element][10]; verify that rows start
// with the element itself and that zeros only trail (hydrodynamic_module.f90:590-647)
int convert_adjacency(lt_engine* e, const int32_t* adj, int elements, std::vector<int>& out,
                      const char* name) {
  out.resize((size_t)elements * 10);
  for (int i = 0; i < elements; i++) {
    bool zero_seen = false;
    for (int k = 0; k < 10; k++) {
      int v = adj[(size_t)i + (size_t)elements * k];
      if (v < 0 || v > elements) return fail(e, LT_ERR_ARG, "%s(%d,%d)=%d out of range", name, i + 1, k + 1, v);
      if (v == 0) zero_seen = true;
      else if (zero_seen) return fail(e, LT_ERR_ARG, "%s row %d: non-zero entry after a zero", name, i + 1);
      out[(size_t)i * 10 + k] = v;
    }
    if (out[(size_t)i * 10] != i + 1)
      return fail(e, LT_ERR_ARG, "%s row %d does not start with the element itself", name, i + 1);
  }
  return LT_OK;
}

int check_elements(lt_engine* e, const int32_t* E, int elements, int nodes, const char* name) {
  for (size_t i = 0; i < (size_t)elements * 4; i++)
    if (E[i] < 1 || E[i] > nodes) return fail(e, LT_ERR_ARG, "%s node id %d out of range", name, E[i]);
  return LT_OK;
}

int check_particle_elements(lt_engine* e, const int32_t* pe, int numpar, int elements, const char* name) {
  for (int i = 0; i < numpar; i++)
    if (pe[i] < 1 || pe[i] > elements)
      return fail(e, LT_ERR_ARG, "%s(%d)=%d is not a valid element id", name, i + 1, pe[i]);
  return LT_OK;
}

constexpr int RNG_LOG2_S = 20;
constexpr size_t RNG_S = (size_t)1 << RNG_LOG2_S;  // MT19937 stream words per block
constexpr int RNG_JOB_SLOTS = 32, RNG_MAX_POLYS = 512;  // (a polynomial is 2.5 KB; distances are k*P-1 and k*P, k up to a step's blocks / P + 1)

size_t slab_elem(const lt_engine* e) { return e->cfg.slab_storage == LT_SLAB_F32 ? 4 : 8; }

// the salinity field is read by behaviours 4, 5, 7 (behavior_module.f90:233-246) and SaltTempOn
bool needs_salt(const lt_config& c) {
  return c.SaltTempOn || c.Behavior == 4 || c.Behavior == 5 || c.Behavior == 7;
}

void fill_params(lt_engine* e, StepParams& P) {
  memset(&P, 0, sizeof(P));
  const lt_config& c = e->cfg;
  P.par = (double*)e->par.p;
  P.numpar = c.numpar;
  P.P_r_element = (int*)e->pr.p; P.P_u_element = (int*)e->pu.p; P.P_v_element = (int*)e->pv.p;
  P.P_behave = (int*)e->P_behave.p; P.dead = (int*)e->dead.p; P.oob = (int*)e->oob.p;
  P.settle = (int*)e->settle.p; P.endpoly = (int*)e->endpoly.p;
  P.hitBottom = (int*)e->hitBottom.p; P.hitLand = (int*)e->hitLand.p;
  P.err_code = (int*)e->err_code.p;
  P.P_swim = (double*)e->P_swim.p; P.P_pediage = (double*)e->P_pediage.p;
  P.P_deadage = (double*)e->P_deadage.p; P.settletime = (double*)e->settletime.p;
  P.rnode = (const RhoNode*)e->rnode.p; P.unode = (const XYNode*)e->unode.p;
  P.vnode = (const XYNode*)e->vnode.p;
  P.RE = (const int4*)e->RE.p; P.UE = (const int4*)e->UE.p; P.VE = (const int4*)e->VE.p;
  P.rAdj = (const int*)e->rAdj.p; P.uAdj = (const int*)e->uAdj.p; P.vAdj = (const int*)e->vAdj.p;
  P.lev = (const double*)e->lev.p;
  P.us = c.us; P.ws = c.ws; P.Vtransform = c.Vtransform;
  P.hc = (double)c.hc;  // REAL promoted (LTRANS.h:66)
  P.z0 = c.z0;
  P.U = e->U.p; P.V = e->V.p; P.W = e->W.p; P.Kh = e->Kh.p; P.zeta = e->zeta.p;
  P.tb = e->phys[e->t_b]; P.tc = e->phys[e->t_c]; P.tf = e->phys[e->t_f];
  P.idt = c.idt;
  P.Behavior = c.Behavior; P.OpenOceanBoundary = c.OpenOceanBoundary; P.mortality = c.mortality;
  P.settlementon = c.settlementon; P.holesExist = c.holesExist; P.ErrorFlag = c.ErrorFlag;
  P.TrackCollisions = c.TrackCollisions;
  P.swimstart = c.swimstart; P.swimfast = c.swimfast; P.sink = c.sink;
  P.dt = c.dt; P.SaltTempOn = c.SaltTempOn;
  P.Sgradient = c.Sgradient; P.Hswimspeed = c.Hswimspeed; P.Swimdepth = c.Swimdepth;
  P.twistart = c.twistart; P.twiend = c.twiend; P.daylength = c.daylength;
  P.Em = c.Em; P.Kd = c.Kd; P.thresh = c.thresh;
  P.S = e->S.p; P.T = e->T.p;
  P.timer = (double*)e->timer.p; P.P_Sprev = (double*)e->P_Sprev.p; P.P_zprev = (double*)e->P_zprev.p;
  P.bottom = (int*)e->bottom.p; P.P_Salt = (double*)e->P_Salt.p; P.P_Temp = (double*)e->P_Temp.p;
  P.bnd = (const double*)e->bnd.p; P.land = (const int*)e->land.p; P.nbounds = e->nbounds;
  P.seg = e->seg; P.use_seg_index = e->use_seg_index;
  P.bx = (const double*)e->bx.p; P.by = (const double*)e->by.p; P.maxbound = e->maxbound;
  P.hx = (const double*)e->hx.p; P.hy = (const double*)e->hy.p; P.hid = (const int*)e->hid.p;
  P.maxisland = e->maxisland; P.numislands = e->numislands;
  P.pip_main = e->pip_main; P.pip_isl = e->pip_isl; P.use_pip_index = e->use_pip_index;
  P.polys = (const double*)e->polys.p; P.holes = (const double*)e->holes.p;
  P.pedges = e->hab.pedges; P.hedges = e->hab.hedges;
  P.polyspecs = (const int*)e->polyspecs.p; P.holespecs = (const int*)e->holespecs.p;
  P.npoly = e->npoly; P.nhole = e->nhole;
  P.maxbdis = (const double*)e->maxbdis.p; P.maxhdis = (const double*)e->maxhdis.p;
  P.elepolys_ptr = (const int*)e->ep_ptr.p; P.elepolys_ids = (const int*)e->ep_ids.p;
  P.polyholes_ptr = (const int*)e->ph_ptr.p; P.polyholes_ids = (const int*)e->ph_ids.p;
  P.minpolyid = e->hab.minpolyid; P.minholeid = e->hab.minholeid;
  P.HTurbOn = c.HTurbOn; P.VTurbOn = c.VTurbOn; P.ConstantHTurb = c.ConstantHTurb; P.PI = c.PI;
  P.rng_ring = (const uint32_t*)e->rng_ring.p;
  P.rng_mask = e->rng_R ? ((unsigned long long)e->rng_R * RNG_S - 1ull) : 0ull;
  P.rng_draws = (c.HTurbOn ? 4 : 0) + (c.VTurbOn ? 2 * (c.idt / 2) : 0);
  P.rank = (const int*)e->rank.p; P.orig = (const int*)e->orig.p;
  P.expect_own = -1;
  P.vt_tab = (VtRec*)e->vt_tab.p;      // records, then the plane of the listed intervals' tension factors
  P.vt_sig = e->vt_tab.p ? (double*)((char*)e->vt_tab.p + (size_t)4 * e->cfg.ws * (((size_t)e->cfg.numpar + 63) / 64 * 64) * sizeof(VtRec)) : nullptr;
  P.vt_bits = (unsigned long long*)e->vt_bits.p;
  P.vt_dev = (double*)e->vt_dev.p;
  P.slow_list = (int*)e->slow_list.p;
  P.vt_q.S = (double*)e->vt_qS.p; P.vt_q.S1 = (double*)e->vt_qS1.p; P.vt_q.S2 = (double*)e->vt_qS2.p;
  P.vt_q.wl = (unsigned short*)e->vt_qwl.p; P.vt_q.count = (int*)e->vt_qcount.p; P.vt_q.cap = e->vt_qcap;
  P.vt_wl_count = (int*)e->vt_wl.p;
  P.vt_wl_list = e->vt_wl.p ? (int*)e->vt_wl.p + 4 : nullptr;
  P.vt_wl_resume = e->vt_wl.p ? (int*)e->vt_wl.p + 4 + 2 * (size_t)e->cfg.numpar : nullptr;
  P.ele_ok = (unsigned char*)e->ele_ok.p;
  P.FreeSlip = c.FreeSlip ? 1 : 0;
  P.fsR = (const unsigned char*)e->fsR.p; P.fsU = (const unsigned char*)e->fsU.p;
  P.fsV = (const unsigned char*)e->fsV.p; P.fsVU = (const unsigned char*)e->fsVU.p;
  P.block_active = (unsigned int*)e->block_active.p;
  P.ws_d = (double*)e->ws_d.p;
  P.ws_i = (int*)e->ws_i.p;
  P.counters = (unsigned long long*)e->counters.p;
}

// ---- boundary indexes (host, once per lt_set_bounds) --------------------
// uniform grid over the segments' bounding box; the list of cell (cx,cy) is the
// ascending union of the segments overlapping cells (cx..cx+1, cy..cy+1)
void build_seg_index(const double* bnd_x, const double* bnd_y, int nb, SegIndex& S, HostIndex& H) {
  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300, len = 0.0;
  for (int i = 0; i < nb; i++) {
    double x1 = bnd_x[2 * i], x2 = bnd_x[2 * i + 1], y1 = bnd_y[2 * i], y2 = bnd_y[2 * i + 1];
    xmin = std::min(xmin, std::min(x1, x2)); xmax = std::max(xmax, std::max(x1, x2));
    ymin = std::min(ymin, std::min(y1, y2)); ymax = std::max(ymax, std::max(y1, y2));
    len += std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
  }
  if (nb == 0) { xmin = ymin = 0.0; xmax = ymax = 1.0; len = 1.0; }
  double cell = 2.0 * len / std::max(nb, 1);
  if (!(cell > 0.0)) cell = 1.0;
  double wx = std::max(xmax - xmin, cell), wy = std::max(ymax - ymin, cell);
  while ((wx / cell + 1.0) * (wy / cell + 1.0) > 4.0e6) cell *= 2.0;
  S.x0 = xmin; S.y0 = ymin;
  S.inv_cw = 1.0 / cell; S.inv_ch = 1.0 / cell;
  S.ncx = (int)std::floor(wx / cell) + 1;
  S.ncy = (int)std::floor(wy / cell) + 1;
  auto cellx = [&](double x) { double f = std::floor((x - S.x0) * S.inv_cw); return (int)std::min<double>(std::max(f, 0.0), S.ncx - 1); };
  auto celly = [&](double y) { double f = std::floor((y - S.y0) * S.inv_ch); return (int)std::min<double>(std::max(f, 0.0), S.ncy - 1); };
  std::vector<std::vector<int>> cells((size_t)S.ncx * S.ncy);
  for (int i = 0; i < nb; i++) {
    double x1 = bnd_x[2 * i], x2 = bnd_x[2 * i + 1], y1 = bnd_y[2 * i], y2 = bnd_y[2 * i + 1];
    int cx0 = cellx(std::min(x1, x2)), cx1 = cellx(std::max(x1, x2));
    int cy0 = celly(std::min(y1, y2)), cy1 = celly(std::max(y1, y2));
    // a query block (bx,by) covers cells bx..bx+1, by..by+1
    for (int by = std::max(cy0 - 1, 0); by <= cy1; by++)
      for (int bx = std::max(cx0 - 1, 0); bx <= cx1; bx++) cells[(size_t)bx + (size_t)by * S.ncx].push_back(i + 1);
  }
  H.ptr.assign((size_t)S.ncx * S.ncy + 1, 0);
  H.ids.clear();
  for (size_t c = 0; c < cells.size(); c++) {
    auto& v = cells[c];
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    H.ptr[c] = (int)H.ids.size();
    H.ids.insert(H.ids.end(), v.begin(), v.end());
  }
  H.ptr[cells.size()] = (int)H.ids.size();
}

// y-slab edge index of a chain of closed polygons given as consecutive points;
// `same_poly(i)` tells whether points i and i+1 (0-based) belong to one polygon
// The slab index of lt_device.h (PipIndex) for the polygon(s) px, py (edge i: point i -> i+1 where same_poly(i)).
// Returns false when the lists would be unreasonably long (the caller then leaves the index off: full algorithm).
template <typename F>
bool build_pip_index(const double* px, const double* py, const int32_t* polyid, int npts, F same_poly, PipIndex& I, HostIndex& H) {
  H.ys.clear(); H.gidx.clear(); H.ptr.clear(); H.ids.clear(); H.rec.clear(); H.pid.clear();
  std::vector<int> edges;
  for (int i = 0; i + 1 < npts; i++)
    if (same_poly(i)) edges.push_back(i);
  for (int i : edges) { H.ys.push_back(py[i]); H.ys.push_back(py[i + 1]); }
  std::sort(H.ys.begin(), H.ys.end());
  H.ys.erase(std::unique(H.ys.begin(), H.ys.end()), H.ys.end());
  const int m = (int)H.ys.size();
  I.m = m; I.ncell = 1; I.y0 = m ? H.ys[0] : 0.0; I.inv_h = 1.0;
  if (m < 2) {   // no edge with a y-extent: nothing can be crossed
    H.ys.resize(2, 0.0); H.gidx.assign(2, 0); H.ptr.assign(2, 0);
    I.m = 0;
    return true;
  }
  // entries: edge e spans the slabs k0 .. k1-1 between its two ordinates
  std::vector<std::vector<int>> slab((size_t)m - 1);
  size_t total = 0;
  for (int i : edges) {
    const double lo = std::min(py[i], py[i + 1]), hi = std::max(py[i], py[i + 1]);
    const int k0 = (int)(std::lower_bound(H.ys.begin(), H.ys.end(), lo) - H.ys.begin());
    const int k1 = (int)(std::lower_bound(H.ys.begin(), H.ys.end(), hi) - H.ys.begin());
    total += (size_t)(k1 - k0);
    if (total > ((size_t)1 << 24)) return false;
    for (int k = k0; k < k1; k++) slab[(size_t)k].push_back(i + 1);   // (edges visited in ascending i: polygon order)
  }
  H.ptr.assign((size_t)m, 0);
  for (int k = 0; k + 1 < m; k++) {
    H.ptr[(size_t)k] = (int)H.ids.size();
    for (int i1 : slab[(size_t)k]) {   // (i1 = 1-based index of the edge's first point)
      H.ids.push_back(i1);
      H.rec.push_back(px[i1 - 1]); H.rec.push_back(py[i1 - 1]); H.rec.push_back(px[i1]); H.rec.push_back(py[i1]);
      H.pid.push_back(polyid ? polyid[i1 - 1] : 0);
    }
  }
  H.ptr[(size_t)m - 1] = (int)H.ids.size();
  // the lookup grid: cell(y) = floor((y - y0) * inv_h), clamped -- monotone in y, the same expression as pip_slab's
  const int ncell = std::min(1 << 17, std::max(16, 4 * m));
  const double h = (H.ys[(size_t)m - 1] - H.ys[0]) / ncell;
  I.ncell = ncell; I.y0 = H.ys[0]; I.inv_h = 1.0 / h;
  auto cell = [&](double y) { const double f = std::floor((y - I.y0) * I.inv_h); return (f < 0.0) ? 0 : (f >= (double)ncell ? ncell - 1 : (int)f); };
  H.gidx.assign((size_t)ncell + 1, 0);
  {
    int j = 0;
    for (int c = 0; c <= ncell; c++) {   // gidx[c] = number of ys whose cell is below c
      while (j < m && cell(H.ys[(size_t)j]) < c) j++;
      H.gidx[(size_t)c] = j;
    }
  }
  return true;
}

template <typename SlabT>
int scatter_field(lt_engine* e, const double* src, DevBuf& dst, int nodes, int L, int slot, int levfast = 0) {
  size_t total = (size_t)nodes * L;
  HIPCHK(e, hipMemcpyAsync(e->staging.p, src, total * 8, hipMemcpyHostToDevice, e->stream));
  unsigned blocks = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL((k_slab_scatter<SlabT>), dim3(blocks), dim3(256), 0, e->stream,
                     (const double*)e->staging.p, (SlabT*)dst.p, nodes, L, e->phys[slot], (int*)e->flags.p, levfast);
  HIPCHK(e, hipGetLastError());
  return LT_OK;
}

template <typename SlabT>
int set_slab_t(lt_engine* e, int slot, const lt_slab* s) {
  const int us = e->cfg.us, ws = e->cfg.ws;
  HIPCHK(e, hipMemcpyAsync(e->staging.p, s->zeta, (size_t)e->rho_nodes * 8, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL((k_zeta_scatter<SlabT>), dim3((e->rho_nodes + 255) / 256), dim3(256), 0, e->stream,
                     (const double*)e->staging.p, (SlabT*)e->zeta.p, e->rho_nodes, e->phys[slot], (int*)e->flags.p);
  HIPCHK(e, hipGetLastError());
  TRY(scatter_field<SlabT>(e, s->Uvel, e->U, e->u_nodes, us, slot, LT_LAYOUT_LEVFAST));
  TRY(scatter_field<SlabT>(e, s->Vvel, e->V, e->v_nodes, us, slot, LT_LAYOUT_LEVFAST));
  TRY(scatter_field<SlabT>(e, s->Wvel, e->W, e->rho_nodes, ws, slot, LT_LAYOUT_LEVFAST));
  if (s->Kh) TRY(scatter_field<SlabT>(e, s->Kh, e->Kh, e->rho_nodes, ws, slot));
  if (s->salt && e->S.p) TRY(scatter_field<SlabT>(e, s->salt, e->S, e->rho_nodes, us, slot));
  if (s->temp && e->T.p) TRY(scatter_field<SlabT>(e, s->temp, e->T, e->rho_nodes, us, slot));
  int flag = 0;
  HIPCHK(e, hipMemcpyAsync(&flag, e->flags.p, 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));  // the caller's arrays may change after return
  if (flag) {
    (void)hipMemsetAsync(e->flags.p, 0, 16, e->stream);
    return fail(e, LT_ERR_SLAB_NOT_F32,
                "slab value is not exactly representable in the device storage type; "
                "use slab_storage = LT_SLAB_F64");
  }
  return LT_OK;
}

int sync_perm(lt_engine* e);  // the host copy of the internal order, after a device-side re-sort

// upload one per-particle array (cols columns, column-major) in the engine's
// internal order: dst[c*n + i] = src[c*n + perm[i]]
template <typename T>
int put_field(lt_engine* e, DevBuf& buf, const T* src, int cols) {
  if (src) TRY(sync_perm(e));
  const size_t n = (size_t)e->cfg.numpar;
  const size_t bytes = n * cols * sizeof(T);
  if (src) {
    std::vector<T> tmp(n * cols);
    const int* pm = e->perm.data();
    for (int c = 0; c < cols; c++) {
      const T* s = src + (size_t)c * n;
      T* d = tmp.data() + (size_t)c * n;
      for (size_t i = 0; i < n; i++) d[i] = s[pm[i]];
    }
    TRY(dev_alloc(e, buf, bytes));
    if (bytes) HIPCHK(e, hipMemcpyAsync(buf.p, tmp.data(), bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));  // tmp goes out of scope
  } else if (!buf.p) {
    TRY(dev_alloc(e, buf, bytes));
    HIPCHK(e, hipMemsetAsync(buf.p, 0, buf.bytes, e->stream));
  }
  return LT_OK;
}

// Download of per-particle arrays in the reference's order (lt_get_particles, the endfile.csv / restart path of
// LTRANS.f90:632-658): each array is put into the reference's order ON THE DEVICE (k_unpermute through `orig`)
// into its part of one staging buffer and copied to the caller's memory from there; nothing waits until
// get_fields_end.  (Round 3 copied each array to a pageable temporary, waited, and scattered it on the host:
// 100 ms for the 240 MB of a million particles.)  The copies land in a pinned buffer of the library, kept between
// calls, and up to eight host threads move each array on to the caller's memory while the next arrives.
// Environment, for measurements: LT_GET_THREADS=n (host threads), LT_GET_NOPIN=1 (copy straight to the caller's
// pageable memory), LT_GET_TIMING=1 (one line on stderr per download).
int get_fields_begin(lt_engine* e, size_t total_bytes) {
  TRY(dev_alloc(e, e->get_stage, total_bytes + 256));
  e->get_stage_used = 0;
  e->get_items.clear();
  const bool nopin = getenv("LT_GET_NOPIN") != nullptr;
  if (e->get_pin_bytes < total_bytes + 256 && !nopin) {
    if (e->get_pin) (void)hipHostFree(e->get_pin);
    e->get_pin = nullptr; e->get_pin_bytes = 0;
    // (kept for the next download; without it -- no pinned memory to be had -- the copies go to the caller's
    // memory directly, at the rate of a pageable destination)
    if (hipHostMalloc(&e->get_pin, total_bytes + 256, hipHostMallocDefault) == hipSuccess) e->get_pin_bytes = total_bytes + 256;
    else { e->get_pin = nullptr; (void)hipGetLastError(); }
  }
  // decided once per download: the pinned path only with a buffer that holds ALL of it (a buffer left by a smaller
  // download stays where it is when LT_GET_NOPIN is set afterwards -- and is not written to)
  e->get_use_pin = !nopin && e->get_pin && e->get_pin_bytes >= total_bytes + 256;
  return LT_OK;
}
template <typename T>
int get_field(lt_engine* e, const DevBuf& buf, T* dst, int cols) {
  if (!dst || !buf.p) return LT_OK;
  const size_t n = (size_t)e->cfg.numpar;
  const size_t bytes = n * cols * sizeof(T);
  if (!bytes) return LT_OK;
  if (e->get_stage_used + bytes > e->get_stage.bytes) return fail(e, LT_ERR_STATE, "get_field: staging buffer too small");
  const size_t off = e->get_stage_used;
  T* stage = (T*)((char*)e->get_stage.p + off);
  e->get_stage_used += (bytes + 255) & ~(size_t)255;
  hipLaunchKernelGGL((ltd::k_unpermute<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, (const T*)buf.p, stage,
                     (const int*)e->orig.p, (int)n, cols);
  if (!e->get_use_pin) {
    HIPCHK(e, hipMemcpyAsync(dst, stage, bytes, hipMemcpyDeviceToHost, e->stream));
    return LT_OK;
  }
  char* pin = (char*)e->get_pin + off;
  HIPCHK(e, hipMemcpyAsync(pin, stage, bytes, hipMemcpyDeviceToHost, e->stream));
  const size_t k = e->get_items.size();
  if (e->get_ev.size() <= k) {
    hipEvent_t ev;
    HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    e->get_ev.push_back(ev);
  }
  HIPCHK(e, hipEventRecord(e->get_ev[k], e->stream));
  e->get_items.push_back({pin, dst, bytes});
  return LT_OK;
}
int get_fields_end(lt_engine* e) {
  HIPCHK(e, hipGetLastError());
  const size_t ni = e->get_items.size();
  const bool tim = getenv("LT_GET_TIMING") != nullptr;
  auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
  const double t_begin = now();
  double t_wait = 0.0, t_copy = 0.0;
  if (ni) {
    // worker t copies the t-th slice of every array, as soon as the array has arrived
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::max(1u, std::min(8u, hw ? hw : 1u));
    if (getenv("LT_GET_THREADS")) nt = std::min(32, std::max(1, atoi(getenv("LT_GET_THREADS"))));
    size_t total = 0;
    for (const auto& it : e->get_items) total += it.bytes;
    if (total < ((size_t)8 << 20)) nt = 1;   // (a small download -- the salinity / temperature columns -- is not worth a thread)
    std::atomic<size_t> ready{0};
    std::atomic<bool> abort_{false};
    auto work = [&](int t) {
      for (size_t k = 0; k < ni; k++) {
        while (ready.load(std::memory_order_acquire) <= k) {
          if (abort_.load(std::memory_order_relaxed)) return;
          std::this_thread::yield();
        }
        const lt_engine::GetItem& it = e->get_items[k];
        const size_t per = ((it.bytes + nt - 1) / nt + 63) & ~(size_t)63;
        const size_t a = std::min(it.bytes, per * (size_t)t), b = std::min(it.bytes, a + per);
        if (b > a) memcpy((char*)it.dst + a, (const char*)it.src + a, b - a);
      }
    };
    // (no exception may cross the C ABI: a thread that cannot be started -- EAGAIN under a process / thread limit --
    // leaves its slices, and those of the ones after it, to this thread; the workers started so far keep theirs)
    std::vector<std::thread> th;
    int started = 1;
    try {
      th.reserve((size_t)nt);
      for (int t = 1; t < nt; t++) { th.emplace_back(work, t); started = t + 1; }
    } catch (...) { }
    hipError_t err = hipSuccess;
    for (size_t k = 0; k < ni && err == hipSuccess; k++) {
      const double t0 = now();
      err = hipEventSynchronize(e->get_ev[k]);
      const double t1 = now();
      if (err == hipSuccess) {
        ready.store(k + 1, std::memory_order_release);
        const lt_engine::GetItem& it = e->get_items[k];   // (this thread takes slice 0 of the array that has just arrived)
        const size_t per = ((it.bytes + nt - 1) / nt + 63) & ~(size_t)63;
        memcpy(it.dst, it.src, std::min(it.bytes, per));
        const size_t a = std::min(it.bytes, per * (size_t)started);   // the slices of workers that could not be started
        if (it.bytes > a) memcpy((char*)it.dst + a, (const char*)it.src + a, it.bytes - a);
      }
      t_wait += t1 - t0; t_copy += now() - t1;
    }
    if (err != hipSuccess) abort_.store(true);
    const double t_j = now();
    for (auto& x : th) x.join();
    if (tim) fprintf(stderr, "get_fields_end: %zu arrays, %d threads: waiting for arrivals %.2f ms, own copies %.2f ms, join %.2f ms, total %.2f ms\n",
                     ni, nt, t_wait, t_copy, now() - t_j, now() - t_begin);
    e->get_items.clear();
    HIPCHK(e, err);
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return LT_OK;
}

// Re-sort of the internal particle order by rho element, on the device.  The order particles are
// held in has no effect on results (every array of the ABI is exchanged in the reference's order
// through `orig`, the random stream is indexed by the reference's particle number), but the gathers
// of a wave stay local only while its 64 particles are in neighbouring elements: with fully mixed
// particles a step takes 2.8x as long.  Stable radix sort of (element, slot), then every per-particle
// array is gathered through the resulting index.
int sync_perm(lt_engine* e) {
  if (!e->perm_stale) return LT_OK;
  HIPCHK(e, hipMemcpyAsync(e->perm.data(), e->orig.p, e->perm.size() * 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->perm_stale = false;
  return LT_OK;
}

int resort(lt_engine* e) {
  const int n = e->cfg.numpar;
  if (n <= 1 || !e->pr.p) return LT_OK;
  TRY(dev_alloc(e, e->sort_keys, (size_t)n * 4));
  TRY(dev_alloc(e, e->sort_idx_in, (size_t)n * 4));
  TRY(dev_alloc(e, e->sort_idx_out, (size_t)n * 4));
  TRY(dev_alloc(e, e->sort_keys_in, (size_t)n * 4));
  const int bits = e->tile_key_bits;
  if (!e->resort_ev[0]) {
    HIPCHK(e, hipEventCreate(&e->resort_ev[0]));
    HIPCHK(e, hipEventCreate(&e->resort_ev[1]));
  }
  HIPCHK(e, hipEventRecord(e->resort_ev[0], e->stream));
  hipLaunchKernelGGL(ltd::k_tile_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, (const int*)e->pr.p,
                     (const int*)e->tile_key.p, (int*)e->sort_keys_in.p, n);
  size_t tmp_bytes = 0;
  HIPCHK(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const int*)e->sort_keys_in.p, (int*)e->sort_keys.p,
                                               (const int*)e->sort_idx_in.p, (int*)e->sort_idx_out.p, n, 0, bits,
                                               e->stream));
  if (tmp_bytes > e->sort_tmp.bytes) TRY(dev_alloc(e, e->sort_tmp, tmp_bytes));
  hipLaunchKernelGGL(ltd::k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, (int*)e->sort_idx_in.p, n);
  HIPCHK(e, hipcub::DeviceRadixSort::SortPairs(e->sort_tmp.p, tmp_bytes, (const int*)e->sort_keys_in.p, (int*)e->sort_keys.p,
                                               (const int*)e->sort_idx_in.p, (int*)e->sort_idx_out.p, n, 0, bits,
                                               e->stream));
  {
    // every per-particle array through the sorted index, in one launch into one staging buffer, and back in another
    ltd::PermList L;
    L.nd = 0;
    size_t total = 0;
    bool overflow = false;
    auto add = [&](DevBuf& b, int elem, int cols) {
      if (!b.p) return;
      if (L.nd >= LT_PERM_MAX) { overflow = true; return; }
      L.d[L.nd].src = b.p; L.d[L.nd].dst = nullptr; L.d[L.nd].elem = elem; L.d[L.nd].cols = cols;
      total += (((size_t)n * elem * cols) + 255) & ~(size_t)255;
      L.nd++;
    };
    add(e->par, 8, 13);
    for (DevBuf* b : {&e->pr, &e->pu, &e->pv, &e->P_behave, &e->dead, &e->oob, &e->settle, &e->endpoly, &e->hitBottom,
                      &e->hitLand, &e->err_code, &e->bottom, &e->orig})
      add(*b, 4, 1);
    add(e->P_swim, 8, 3);
    for (DevBuf* b : {&e->P_pediage, &e->P_deadage, &e->settletime, &e->timer, &e->P_Sprev, &e->P_zprev, &e->P_Salt,
                      &e->P_Temp})
      add(*b, 8, 1);
    if (overflow) return fail(e, LT_ERR_STATE, "resort: more per-particle arrays than LT_PERM_MAX");
    TRY(dev_alloc(e, e->permute_tmp, total + 256));
    size_t off = 0;
    for (int k = 0; k < L.nd; k++) {
      L.d[k].dst = (char*)e->permute_tmp.p + off;
      off += (((size_t)n * L.d[k].elem * L.d[k].cols) + 255) & ~(size_t)255;
    }
    const unsigned pb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ltd::k_permute_multi, dim3(pb), dim3(256), 0, e->stream, L, (const int*)e->sort_idx_out.p, n);
    hipLaunchKernelGGL(ltd::k_copy_back_multi, dim3(pb), dim3(256), 0, e->stream, L, n);
  }
  // "this particle's stored elements are verified" is per slot: verify again
  if (e->ele_ok.p) HIPCHK(e, hipMemsetAsync(e->ele_ok.p, 0, e->ele_ok.bytes, e->stream));
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipEventRecord(e->resort_ev[1], e->stream));
  e->resorts += 1;
  e->perm_stale = true;
  return LT_OK;
}

// ---- device MT19937 stream --------------------------------------------------

const ltmt::Poly& mt_phi() {
  static const ltmt::Poly phi = ltmt::characteristic_polynomial();
  return phi;
}

// (the attribute is per device: every engine sets it on its own device, after hipSetDevice)
int rng_configure_kernels(lt_engine* e) {
  if (!e->rng_kernels_configured) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipFuncSetAttribute((const void*)ltmt::k_mt_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ltmt::LDS_BYTES * ltmt::GEN_PER_WG)));
    HIPCHK(e, hipFuncSetAttribute((const void*)ltmt::k_mt_jump, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ltmt::LDS_BYTES));
    HIPCHK(e, hipFuncSetAttribute((const void*)ltmt::k_mt_jump_each, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ltmt::LDS_BYTES));
    HIPCHK(e, hipFuncSetAttribute((const void*)ltmt::k_mt_jump_fan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ltmt::LDS_BYTES));
    e->rng_kernels_configured = true;
  }
  return LT_OK;
}

// position the P generator states: state p at the start of block p (p >= 1), state 0
// = the given mt array (word 624; block 0 starts with those words themselves)
int rng_init_device(lt_engine* e) {
  TRY(rng_configure_kernels(e));
  const int P = e->rng_P;
  const ltmt::Poly& phi = mt_phi();
  if (phi.empty()) return fail(e, LT_ERR_DEVICE, "MT19937 characteristic polynomial failed");
  e->rng_g0 = ltmt::jump_polynomial(phi, RNG_S - 624);
  e->rng_gS = ltmt::jump_polynomial(phi, RNG_S);
  // State 1 = the mt array moved to the start of block 1; then by doubling: with the states 1 .. h at the starts
  // of blocks 1 .. h, ONE launch of the jump over h blocks places the next h of them (x^(2h*S) = the square of
  // x^(h*S): one modular product on the host per round, beside the device's work).  log2(P) launches -- a launch
  // per state, each waiting for the one before, was 169 ms of a turbulence engine's start-up at P = 256.
  std::vector<std::vector<uint32_t>> pw;  // x^(S), x^(2S), x^(4S), ...
  pw.push_back(e->rng_gS);
  for (int h = 2; h < P - 1; h *= 2) pw.push_back(ltmt::poly_mulmod(phi, pw.back(), pw.back()));
  std::vector<uint32_t> flat((size_t)(pw.size() + 1) * ltmt::GW);
  std::copy(e->rng_g0.begin(), e->rng_g0.end(), flat.begin());
  for (size_t r = 0; r < pw.size(); r++) std::copy(pw[r].begin(), pw[r].end(), flat.begin() + (r + 1) * ltmt::GW);
  DevBuf d_g;
  TRY(upload(e, d_g, flat.data(), flat.size() * 4));
  if (!e->rng_polys.p) TRY(dev_alloc(e, e->rng_polys, (size_t)RNG_MAX_POLYS * ltmt::GW * 4));
  e->rng_poly_slot.clear();
  e->rng_family_done = false;
  e->rng_rebased = false;
  if (!e->rng_states.p) TRY(dev_alloc(e, e->rng_states, (size_t)P * 624 * 4));
  if (!e->rng_jobs) HIPCHK(e, hipHostMalloc((void**)&e->rng_jobs, sizeof(ltmt::BlockJob) * (size_t)RNG_JOB_SLOTS * P, hipHostMallocMapped));
  HIPCHK(e, hipMemcpyAsync(e->rng_states.p, e->mt, 624 * 4, hipMemcpyHostToDevice, e->stream));
  if (P > 1)
    hipLaunchKernelGGL(ltmt::k_mt_jump_fan, dim3(1), dim3(ltmt::JUMP_WIDE), ltmt::LDS_BYTES, e->stream,
                       (uint32_t*)e->rng_states.p, 0, 1, (const uint32_t*)d_g.p);
  {
    int have = 1, r = 0;  // states 1 .. have are in place; the next round jumps over `have` blocks
    while (have < P - 1) {
      const int cnt = std::min(have, P - 1 - have);
      hipLaunchKernelGGL(ltmt::k_mt_jump_fan, dim3((unsigned)cnt), dim3(ltmt::JUMP_WIDE), ltmt::LDS_BYTES, e->stream,
                         (uint32_t*)e->rng_states.p, 1, have + 1, (const uint32_t*)d_g.p + (size_t)(r + 1) * ltmt::GW);
      have += cnt;
      r += 1;
    }
  }
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipStreamSynchronize(e->stream));
  (void)hipFree(d_g.p);
  e->rng_pos.assign(P, 0);
  for (int p = 0; p < P; p++) e->rng_pos[p] = p;
  e->rng_gen0_raw = true;
  e->rng_ring_blk.assign((size_t)e->rng_R, -1);
  e->rng_cur = (uint64_t)e->mti;
  e->rng_ready = true;
  return LT_OK;
}

// x^(d*S) mod phi on the host.  The distances that occur are j*P - 1 (a state's turn came again after j rounds)
// and j*P (from its first position): products of x^(P*S), kept by power, with x^((P-1)*S) -- one or two modular
// multiplications (~8 ms each) instead of the 50 ms of a square-and-multiply from scratch.
const std::vector<uint32_t>& rng_pow_round(lt_engine* e, long long j) {  // x^(j*P*S), j >= 1
  if (e->rng_pow_PS.empty()) {
    e->rng_gPm1 = ltmt::jump_polynomial(mt_phi(), (uint64_t)(e->rng_P - 1) * RNG_S);
    e->rng_pow_PS.push_back(ltmt::poly_mulmod(mt_phi(), e->rng_gPm1, e->rng_gS));
  }
  while ((long long)e->rng_pow_PS.size() < j)
    e->rng_pow_PS.push_back(ltmt::poly_mulmod(mt_phi(), e->rng_pow_PS.back(), e->rng_pow_PS[0]));
  return e->rng_pow_PS[(size_t)j - 1];
}
std::vector<uint32_t> rng_jump_poly(lt_engine* e, long long d) {
  const long long P = e->rng_P;
  if (d >= P - 1 && d <= 64 * P) {
    if (d % P == 0) return rng_pow_round(e, d / P);
    if ((d + 1) % P == 0) {
      const long long j = (d + 1) / P;
      (void)rng_pow_round(e, 1);
      return j == 1 ? e->rng_gPm1 : ltmt::poly_mulmod(mt_phi(), rng_pow_round(e, j - 1), e->rng_gPm1);
    }
  }
  return ltmt::jump_polynomial(mt_phi(), (uint64_t)d * RNG_S);
}
// the slot of the jump polynomial of d blocks (computed at its first use and kept)
int rng_poly_for(lt_engine* e, long long d, int* slot) {
  auto it = e->rng_poly_slot.find(d);
  if (it != e->rng_poly_slot.end()) { *slot = it->second; return LT_OK; }
  const int k = (int)e->rng_poly_slot.size();
  if (k >= RNG_MAX_POLYS) return fail(e, LT_ERR_STATE, "MT19937 stream: more than %d distinct jump distances", RNG_MAX_POLYS);
  const std::vector<uint32_t> g = rng_jump_poly(e, d);
  HIPCHK(e, hipMemcpy((uint32_t*)e->rng_polys.p + (size_t)k * ltmt::GW, g.data(), (size_t)ltmt::GW * 4, hipMemcpyHostToDevice));
  e->rng_poly_slot[d] = k;
  *slot = k;
  return LT_OK;
}
// A shard's first jump over other shards' words (j*P - 1 blocks, j > 1) says what the later ones will be: the
// same state comes back after j-1, j or j+1 rounds, or twice that when a window misses it once.  Those are
// computed with it, so that no step after the first two pays for a polynomial.
int rng_prefetch_family(lt_engine* e, long long d) {
  const long long P = e->rng_P;
  if ((d + 1) % P != 0 || d < 2 * P - 1 || e->rng_family_done) return LT_OK;
  e->rng_family_done = true;
  const long long j = (d + 1) / P;
  for (long long m : {j - 1, j, j + 1, 2 * j - 1, 2 * j, 2 * j + 1, 2 * j + 2}) {
    int slot;
    if (m >= 1 && m <= 64) TRY(rng_poly_for(e, m * P - 1, &slot));
  }
  return LT_OK;
}

// make stream words [first, last] resident in the ring.  Block b belongs to generator state b mod P, which is
// positioned at the start of the block it wrote last + 1 -- or further back: a block nobody asks for is never
// generated.  A single engine asks for every block in turn, so its states jump over the P-1 blocks of the others
// after every block written.  A SHARD's window (lt_set_rng_window*) touches a few blocks of every step's part
// of the stream: the blocks of the other shards in between are skipped, and the owner of a needed block jumps
// k*P - 1 blocks, k = the number of turns it was not asked -- a handful of distinct distances whatever the
// windows do, one polynomial each (rng_poly_for).  (Before: a rank of the 8-rank run of BASELINE configs[3]
// generated all 1183 blocks of a step to keep 150 of them.)
int rng_ensure(lt_engine* e, uint64_t first, uint64_t last, hipStream_t gen_stream = nullptr) {
  if (!gen_stream) gen_stream = e->stream;
  if (!e->rng_ready) TRY(rng_init_device(e));
  const int P = e->rng_P;
  const long long b_lo = std::max<long long>((long long)(first >> RNG_LOG2_S) - 1, 0);  // keep one block of history
  const long long b_hi = (long long)(last >> RNG_LOG2_S);
  const long long need = b_hi - b_lo + 2;
  int R = 4;
  while (R < need) R <<= 1;
  if (R > e->rng_R) {
    // (re)allocate the ring; resident blocks still needed are copied into the larger ring
    DevBuf nr;
    TRY(dev_alloc(e, nr, (size_t)R * RNG_S * 4));
    std::vector<long long> nblk((size_t)R, -1);
    if (e->rng_ring.p) {
      for (int sl = 0; sl < e->rng_R; sl++) {
        const long long b = e->rng_ring_blk[sl];
        if (b < b_lo || b > b_hi) continue;
        HIPCHK(e, hipMemcpyAsync((uint32_t*)nr.p + (size_t)(b % R) * RNG_S, (uint32_t*)e->rng_ring.p + (size_t)sl * RNG_S,
                                 RNG_S * 4, hipMemcpyDeviceToDevice, e->stream));
        nblk[(size_t)(b % R)] = b;
      }
      HIPCHK(e, hipStreamSynchronize(e->stream));
      (void)hipFree(e->rng_ring.p);
    }
    e->rng_ring = nr;
    e->rng_R = R;
    e->rng_ring_blk.swap(nblk);
  }
  std::vector<long long> todo;
  for (long long b = b_lo; b <= b_hi; b++)
    if (e->rng_ring_blk[(size_t)(b % e->rng_R)] != b) todo.push_back(b);
  if (todo.empty()) return LT_OK;
  if (e->rng_gen0_raw && todo.front() > 0) {
    // block 0 is not going to be written by this engine: state 0 moves from the mt array (word 624) to block P,
    // its next turn (x^(S-624) to block 1, then P-1 blocks)
    DevBuf d_g0;
    TRY(upload(e, d_g0, e->rng_g0.data(), e->rng_g0.size() * 4));
    hipLaunchKernelGGL(ltmt::k_mt_jump, dim3(1), dim3(256), ltmt::LDS_BYTES, e->stream, (uint32_t*)e->rng_states.p, 0, 0,
                       (const uint32_t*)d_g0.p);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    (void)hipFree(d_g0.p);
    e->rng_pos[0] = 1;
    e->rng_gen0_raw = false;
  }
  if (!e->rng_chain_done && todo.front() >= (long long)P) {
    // a window that does not start where the stream starts: this engine is a shard.  Its states will be asked
    // after ever longer pauses until each has had its first turn (j*P blocks from where it sits, j growing by
    // a step's rounds per step): the powers of x^(P*S) are formed now (~0.4 s, once) instead of a few in each
    // of the first steps.
    (void)rng_pow_round(e, 48);
    e->rng_chain_done = true;
  }
  if (!e->rng_rebased && todo.front() >= 2 * (long long)P) {
    // A shard whose window starts far into the stream: the whole pool moves j0 rounds ahead at once (one
    // polynomial, one launch), so that afterwards every distance is a small j*P or j*P - 1 -- otherwise each
    // state would make its first jump from its initial position, a longer one every step.
    bool fresh = true;
    for (int p = 0; p < P; p++) fresh = fresh && e->rng_pos[(size_t)p] == (p == 0 ? 1 : p);
    if (fresh) {
      const long long j0 = todo.front() / P - 1;   // (every state stays behind the first block it owns in the window)
      int slot;
      TRY(rng_poly_for(e, j0 * P, &slot));
      hipLaunchKernelGGL(ltmt::k_mt_jump_each, dim3((unsigned)P), dim3(256), ltmt::LDS_BYTES, e->stream,
                         (uint32_t*)e->rng_states.p, (const uint32_t*)e->rng_polys.p + (size_t)slot * ltmt::GW);
      HIPCHK(e, hipGetLastError());
      HIPCHK(e, hipStreamSynchronize(e->stream));
      for (int p = 0; p < P; p++) e->rng_pos[(size_t)p] += j0 * P;
    }
    e->rng_rebased = true;
  }
  int launches = 0;
  for (size_t next = 0; next < todo.size();) {
    if (launches == RNG_JOB_SLOTS) {  // every job slot is in flight
      HIPCHK(e, hipStreamSynchronize(gen_stream));
      launches = 0;
    }
    // (a job slot is host memory the launch reads when it RUNS: before it is written again, the launch that
    // last used it must have finished -- the host may be many steps ahead of the device)
    if (e->rng_job_ev.empty()) {
      e->rng_job_ev.assign(RNG_JOB_SLOTS, nullptr);
      for (auto& ev : e->rng_job_ev) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int job_slot = e->rng_job_slot;
    if (hipEventQuery(e->rng_job_ev[(size_t)job_slot]) != hipSuccess)   // (32 launches back: done, unless the host runs far ahead)
      HIPCHK(e, hipEventSynchronize(e->rng_job_ev[(size_t)job_slot]));
    ltmt::BlockJob* jobs = e->rng_jobs + (size_t)job_slot * P;
    e->rng_job_slot = (e->rng_job_slot + 1) % RNG_JOB_SLOTS;
    // one launch: the next blocks as long as their owners differ (P consecutive blocks at most)
    int nj_total = 0;
    std::vector<char> used((size_t)P, 0);
    for (; next < todo.size(); next++) {
      const long long blk = todo[next];
      const int p = (int)(blk % P);
      if (used[(size_t)p]) break;
      used[(size_t)p] = 1;
      const long long d = blk - e->rng_pos[(size_t)p];
      if (d < 0) return fail(e, LT_ERR_STATE, "MT19937 stream: block %lld lies behind its generator state", blk);
      int slot = -1;
      if (d > 0) {
        TRY(rng_poly_for(e, d, &slot));
        TRY(rng_prefetch_family(e, d));
      }
      ltmt::BlockJob& jb = jobs[nj_total++];
      jb.blk = blk; jb.state = p; jb.poly = slot; jb.store = 1; jb.pad_ = 0;
      e->rng_pos[(size_t)p] = blk + 1;
      if (p == 0) e->rng_gen0_raw = false;
      e->rng_ring_blk[(size_t)(blk % e->rng_R)] = blk;
    }
    ltmt::BlockJob* djobs = nullptr;
    HIPCHK(e, hipHostGetDevicePointer((void**)&djobs, jobs, 0));
    // (block 0 -- it starts with the words of the mt array -- goes alone: its loop counts differ)
    for (int j0 = 0; j0 < nj_total;) {
      const int nj = (jobs[j0].blk == 0) ? 1 : nj_total - j0;
      const unsigned wgs = (unsigned)((nj + ltmt::GEN_PER_WG - 1) / ltmt::GEN_PER_WG);
      hipLaunchKernelGGL(ltmt::k_mt_block, dim3(wgs), dim3(ltmt::GEN_THREADS * ltmt::GEN_PER_WG),
                         ltmt::LDS_BYTES * ltmt::GEN_PER_WG, gen_stream, (uint32_t*)e->rng_states.p, (uint32_t*)e->rng_ring.p,
                         (const ltmt::BlockJob*)(djobs + j0), nj, e->rng_R, (const uint32_t*)e->rng_polys.p, (int)RNG_S);
      j0 += nj;
    }
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(e->rng_job_ev[(size_t)job_slot], gen_stream));
    launches++;
  }
  return LT_OK;
}

// the first consumer of the step's random words waits for the generators
void wait_rng(lt_engine* e) {
  if (e->rng_pending) (void)hipStreamWaitEvent(e->stream, e->rng_done, 0);
  e->rng_pending = false;
}
// VTurb after k_rk.  What depends on what: k_vt_sigs on the table; k_vt_norm on the step's random words only;
// k_vt_walk_fast on the table and the deviates of ITS particles -- not on k_vt_sigs: a listed interval carries a
// non-zero mark from the table kernel on (VT_SIG_PENDING) and a walk stops there --; k_vt_walk on all of them.
// k_vt_sigs and the walks wait for their own chains of dependent operations and k_vt_norm is bound by its instruction
// stream, so they are worth running beside each other; each is small in registers (38 .. 96), unlike k_rk and
// k_vt_table, which fill the register file with two waves per SIMD and share it with nobody.
// (in two parts: the head needs none of the step's random words and not the shard's place in the stream, so lt_step
// can enqueue it before the host knows the step's draw totals)
int launch_vturb_head(lt_engine* e, const StepParams& P, unsigned blocks) {
  // (k_vt_table, k_vt_norm and k_vt_walk_fast live in lt_rk_kernel.hip: lt_rk_launch.h)
  launch_k_vt_table(e->cfg.slab_storage == LT_SLAB_F32, e->cfg.Vtransform, P.fastdiv != 0, blocks, e->stream, P);
  if (e->vt_aside && !e->vt_stream) {
    HIPCHK(e, hipStreamCreateWithFlags(&e->vt_stream, hipStreamNonBlocking));
    for (auto& ev : e->vt_ev) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  hipStream_t s_sigs = e->vt_aside ? e->vt_stream : e->stream;
  if (e->vt_aside) {
    HIPCHK(e, hipEventRecord(e->vt_ev[0], e->stream));            // the table is there
    HIPCHK(e, hipStreamWaitEvent(s_sigs, e->vt_ev[0], 0));
  }
  hipLaunchKernelGGL(k_vt_sigs, dim3(blocks), dim3(256), 0, s_sigs, P);
  if (e->vt_aside) HIPCHK(e, hipEventRecord(e->vt_ev[1], s_sigs));
  return LT_OK;
}
int launch_vturb_tail(lt_engine* e, const StepParams& P, unsigned blocks) {
  wait_rng(e);
  launch_k_vt_norm(blocks, e->stream, P);
  launch_k_vt_walk_fast(blocks, e->stream, P);
  if (e->vt_aside) HIPCHK(e, hipStreamWaitEvent(e->stream, e->vt_ev[1], 0));
  // the walks that met a tension factor (or a SigErr), the rest of the walk with every form: as many workgroups as
  // could be needed (a stop and a SigErr entry per particle), the ones beyond the list's end leave at once
  const unsigned walk_blocks = (unsigned)((2 * (size_t)P.numpar + LT_VT_WALK_BLOCK - 1) / LT_VT_WALK_BLOCK);
  hipLaunchKernelGGL(k_vt_walk, dim3(walk_blocks), dim3(LT_VT_WALK_BLOCK), 0, e->stream, P);
  return LT_OK;
}
// k_rk is instantiated per slab type and per Vtransform (the s-level formula is then straight-line code in
// the hot loop), in lt_rk_kernel.hip
void launch_rk(lt_engine* e, const StepParams& P, unsigned blocks) {
  // (a multiple of 8 workgroups: k_rk maps them to particle chunks per XCD)
#if defined(LT_RK_XCD_CHUNK) && LT_RK_XCD_CHUNK > 0
  const unsigned rkunit = 8u * LT_RK_XCD_CHUNK;
  const unsigned rkblocks = (((unsigned)(((size_t)blocks * 256 + LT_RK_BLOCK - 1) / LT_RK_BLOCK) + rkunit - 1u) / rkunit) * rkunit;
#else
  const unsigned rkblocks = ((unsigned)(((size_t)blocks * 256 + LT_RK_BLOCK - 1) / LT_RK_BLOCK) + 7u) & ~7u;
#endif
  static const unsigned dynlds = getenv("LT_RK_DYNLDS") ? (unsigned)atoi(getenv("LT_RK_DYNLDS")) : 0u;  // occupancy experiments
  launch_k_rk(e->cfg.slab_storage == LT_SLAB_F32, e->cfg.Vtransform, e->cfg.FreeSlip != 0, rkblocks, dynlds, e->stream, P);
}

}  // namespace

namespace {
struct RomField {
  const void* src; const int* mask; DevBuf* dst;
  int n1, n2, L;           // array extents (i, j, level)
  int i1, i2, j1, j2;      // window, 1-based inclusive
  size_t off;              // element offset of the packed window in the staging buffers
  int levfast;             // destination layout (lt_device.h)
  size_t count() const { return (size_t)(i2 - i1 + 1) * (j2 - j1 + 1) * L; }
};

// copy the window of one field into the packed staging layout [level][j][i]
void pack_field(const RomField& f, size_t es, char* pin, int k_lo, int k_hi) {
  const size_t wi = (size_t)(f.i2 - f.i1 + 1), wj = (size_t)(f.j2 - f.j1 + 1);
  const char* src = (const char*)f.src;
  char* dst = pin + f.off * es;
  for (int k = k_lo; k < k_hi; k++) {
    if (wi == (size_t)f.n1) {  // full rows: one contiguous block of rows per level
      memcpy(dst + (size_t)k * wi * wj * es,
             src + (((size_t)k * f.n2 + (f.j1 - 1)) * f.n1) * es, wi * wj * es);
    } else {
      for (size_t jj = 0; jj < wj; jj++)
        memcpy(dst + (((size_t)k * wj + jj) * wi) * es,
               src + (((size_t)k * f.n2 + (f.j1 - 1 + jj)) * f.n1 + (f.i1 - 1)) * es, wi * es);
    }
  }
}

template <typename SlabT, typename HostT>
void launch_rom_scatter(lt_engine* e, const RomField& f, int pslot) {
  const size_t total = f.count();
  if (!total) return;
  hipLaunchKernelGGL((k_rom_scatter<SlabT, HostT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     e->copy_stream, (const HostT*)e->stage_rom.p + f.off, (SlabT*)f.dst->p, f.mask, f.n1,
                     f.i1, f.i2 - f.i1 + 1, f.j1, f.j2 - f.j1 + 1, f.L, pslot, (int*)e->flags.p + 1, f.levfast);
}
}  // namespace

extern "C" {

int lt_abi_version(void) { return LT_ABI_VERSION; }

const char* lt_last_error(const lt_engine* e) { return e ? e->err : "null engine"; }

#ifndef LT_SOURCE_HASH
#define LT_SOURCE_HASH "unknown"
#endif
const char* lt_source_hash(void) { return LT_SOURCE_HASH; }

int lt_create(const lt_config* cfg, lt_engine** out) {
  if (!cfg || !out) return LT_ERR_ARG;
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(lt_config) || cfg->abi_version != LT_ABI_VERSION)
    return LT_ERR_ABI;
  if (cfg->numpar < 0 || cfg->us < 5 || cfg->ws != cfg->us + 1 || cfg->ws > LT_LEV_STRIDE ||
      cfg->idt <= 0 || cfg->Vtransform < 1 || cfg->Vtransform > 3)
    return LT_ERR_ARG;
  if (cfg->Behavior < 0 || cfg->Behavior > 7) return LT_ERR_UNSUPPORTED;
  if (cfg->slab_storage != LT_SLAB_F32 && cfg->slab_storage != LT_SLAB_F64) return LT_ERR_ARG;
  {
    // the draw offsets of a shard (prefix sum over its particles) are 32-bit: a shard may hold at most
    // (2^31 - 1) / (4 + 2 (idt/2) + 3) particles when it draws -- 16.9M at idt = 120 (a larger run shards)
    const int64_t per = (cfg->HTurbOn ? 4 : 0) + (cfg->VTurbOn ? 2 * (int64_t)(cfg->idt / 2) : 0) +
                        ((cfg->Behavior >= 1 && cfg->Behavior <= 5) ? 3 : 0);
    if (per > 0 && (int64_t)cfg->numpar * per > (int64_t)INT32_MAX) return LT_ERR_ARG;
  }
  lt_engine* e = new (std::nothrow) lt_engine();
  if (!e) return LT_ERR_NOMEM;
  e->cfg = *cfg;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    delete e;
    return LT_ERR_DEVICE;
  }
  int dev = cfg->device;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  }
  if (dev >= ndev || hipSetDevice(dev) != hipSuccess ||
      hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&e->rng_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->prefetch_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->compute_mark, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->pin_free[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->pin_free[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->rng_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->step_done, hipEventDisableTiming) != hipSuccess) {
    delete e;
    return LT_ERR_DEVICE;
  }
  e->device = dev;
  if (hipMalloc(&e->counters.p, CNT_ALLOC * sizeof(unsigned long long)) != hipSuccess ||
      hipHostMalloc((void**)&e->h_counters, CNT_ALLOC * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&e->flags.p, 16) != hipSuccess) {
    lt_destroy(e);
    return LT_ERR_DEVICE;
  }
  e->counters.bytes = CNT_ALLOC * sizeof(unsigned long long);
  (void)hipMemsetAsync(e->counters.p, 0, e->counters.bytes, e->stream);
  e->flags.bytes = 16;
  (void)hipMemsetAsync(e->flags.p, 0, 16, e->stream);
  const char* envs = getenv("LT_NO_SORT");
  if (envs && atoi(envs) != 0) e->sort_particles = false;
  e->resort_every = (cfg->resort_every == 0) ? 32 : cfg->resort_every;
  const char* envr = getenv("LT_RESORT_EVERY");
  if (envr) e->resort_every = atoi(envr);
  // (tests: a small pool of generator states makes short runs go round it many times)
  const char* envp = getenv("LT_RNG_STATES");
  if (envp && atoi(envp) >= 2 && atoi(envp) <= 1024) e->rng_P = atoi(envp);
  if (getenv("LT_VT_ASIDE")) e->vt_aside = atoi(getenv("LT_VT_ASIDE")) != 0;   // (measurements)
  const char* env = getenv("LT_BRUTE_BOUNDS");
  if (env && atoi(env) != 0) { e->use_seg_index = 0; e->use_pip_index = 0; }
  // init_genrand(seed), random_module.f90:108-119 / LTRANS.f90:241
  e->mt[0] = (uint32_t)cfg->seed;
  for (int i = 1; i < 624; i++) e->mt[i] = 1812433253u * (e->mt[i - 1] ^ (e->mt[i - 1] >> 30)) + (uint32_t)i;
  e->mti = 624;
  *out = e;
  return LT_OK;
}

int lt_destroy(lt_engine* e) {
  if (!e) return LT_ERR_ARG;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
  if (e->rng_stream) (void)hipStreamSynchronize(e->rng_stream);
  DevBuf* bufs[] = {&e->rnode, &e->unode, &e->vnode, &e->RE, &e->UE, &e->VE, &e->rAdj, &e->uAdj,
                    &e->vAdj, &e->lev, &e->fsR, &e->fsU, &e->fsV, &e->fsVU, &e->U, &e->V, &e->W, &e->Kh, &e->zeta, &e->staging, &e->stage_rom, &e->rmask, &e->umask, &e->vmask,
                    &e->flags, &e->bnd, &e->land, &e->bx, &e->by, &e->hx, &e->hy, &e->hid,
                    &e->seg_ptr, &e->seg_ids, &e->pm_ptr, &e->pm_ids, &e->pi_ptr, &e->pi_ids, &e->pm_rec, &e->pi_rec, &e->pi_pid, &e->pm_ys, &e->pi_ys,
                    &e->polys, &e->holes, &e->polyspecs, &e->holespecs, &e->maxbdis, &e->maxhdis,
                    &e->ep_ptr, &e->ep_ids, &e->ph_ptr, &e->ph_ids, &e->par, &e->pr, &e->pu,
                    &e->pv, &e->P_behave, &e->dead, &e->oob, &e->settle, &e->endpoly,
                    &e->hitBottom, &e->hitLand, &e->err_code, &e->P_swim, &e->P_pediage,
                    &e->P_deadage, &e->settletime, &e->timer, &e->P_Sprev, &e->P_zprev, &e->bottom, &e->P_Salt, &e->P_Temp, &e->S, &e->T, &e->counters, &e->ws_d, &e->ws_i, &e->orig, &e->ele_ok, &e->block_active, &e->get_stage, &e->count_slots, &e->out_cols, &e->out_hits, &e->rng_states, &e->rng_ring, &e->rng_polys, &e->flag, &e->rank, &e->scan_tmp,
                    &e->vt_tab, &e->vt_bits, &e->vt_qS, &e->vt_qS1, &e->vt_qS2, &e->vt_qwl, &e->vt_qcount, &e->vt_wl, &e->vt_dev, &e->slow_list, &e->sort_keys, &e->sort_keys_in, &e->tile_key, &e->sort_idx_in, &e->sort_idx_out,
                    &e->sort_tmp, &e->permute_tmp};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->phase_ev) (void)hipEventDestroy(ev);
  if (e->h_counters) (void)hipHostFree(e->h_counters);
  if (e->h_miss) (void)hipHostFree(e->h_miss);
  if (e->miss_ev) (void)hipEventDestroy(e->miss_ev);
  if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
  for (int i = 0; i < 2; i++) {
    if (e->pin[i]) (void)hipHostFree(e->pin[i]);
    if (e->pin_free[i]) (void)hipEventDestroy(e->pin_free[i]);
  }
  if (e->comm) (void)lt_comm_destroy(e);
  for (DevBuf* b : {&e->comm_send, &e->comm_recv, &e->comm_out, &e->comm_counts, &e->cnt_dev})
    if (b->p) (void)hipFree(b->p);
  if (e->h_cnt) (void)hipHostFree(e->h_cnt);
  if (e->cnt_ready) (void)hipEventDestroy(e->cnt_ready);
  if (e->cnt_arrived) (void)hipEventDestroy(e->cnt_arrived);
  if (e->h_own) (void)hipHostFree(e->h_own);
  if (e->own_ev) (void)hipEventDestroy(e->own_ev);
  if (e->prefetch_done) (void)hipEventDestroy(e->prefetch_done);
  if (e->compute_mark) (void)hipEventDestroy(e->compute_mark);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->rng_done) (void)hipEventDestroy(e->rng_done);
  if (e->step_done) (void)hipEventDestroy(e->step_done);
  for (hipEvent_t ev : e->resort_ev) if (ev) (void)hipEventDestroy(ev);
  if (e->comm_packed) (void)hipEventDestroy(e->comm_packed);
  if (e->comm_done) (void)hipEventDestroy(e->comm_done);
  if (e->comm_stream) (void)hipStreamDestroy(e->comm_stream);
  for (hipEvent_t ev : e->rng_job_ev) if (ev) (void)hipEventDestroy(ev);
  if (e->rng_jobs) (void)hipHostFree(e->rng_jobs);
  if (e->rng_stream) (void)hipStreamDestroy(e->rng_stream);
  for (hipEvent_t ev : e->vt_ev) if (ev) (void)hipEventDestroy(ev);
  if (e->vt_stream) { (void)hipStreamSynchronize(e->vt_stream); (void)hipStreamDestroy(e->vt_stream); }
  if (e->get_pin) (void)hipHostFree(e->get_pin);
  for (auto ev : e->get_ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return LT_OK;
}

int lt_set_grid(lt_engine* e, const lt_grid* g) {
  if (!e || !g) return LT_ERR_ARG;
  if (!g->rx || !g->ry || !g->ux || !g->uy || !g->vx || !g->vy || !g->depth || !g->rho_angle ||
      !g->RE || !g->UE || !g->VE || !g->r_Adjacent || !g->u_Adjacent || !g->v_Adjacent ||
      !g->SC || !g->CS || !g->SCW || !g->CSW)
    return fail(e, LT_ERR_ARG, "lt_set_grid: NULL array");
  if (g->rho_nodes <= 0 || g->u_nodes <= 0 || g->v_nodes <= 0 || g->rho_elements <= 0 ||
      g->u_elements <= 0 || g->v_elements <= 0)
    return fail(e, LT_ERR_ARG, "lt_set_grid: bad extents");
  HIPCHK(e, hipSetDevice(e->device));
  TRY(check_elements(e, g->RE, g->rho_elements, g->rho_nodes, "RE"));
  TRY(check_elements(e, g->UE, g->u_elements, g->u_nodes, "UE"));
  TRY(check_elements(e, g->VE, g->v_elements, g->v_nodes, "VE"));
  std::vector<RhoNode> rn(g->rho_nodes);
  for (int i = 0; i < g->rho_nodes; i++) rn[i] = {g->rx[i], g->ry[i], g->depth[i], g->rho_angle[i]};
  std::vector<XYNode> un(g->u_nodes), vn(g->v_nodes);
  for (int i = 0; i < g->u_nodes; i++) un[i] = {g->ux[i], g->uy[i]};
  for (int i = 0; i < g->v_nodes; i++) vn[i] = {g->vx[i], g->vy[i]};
  std::vector<int> ra, ua, va;
  TRY(convert_adjacency(e, g->r_Adjacent, g->rho_elements, ra, "r_Adjacent"));
  TRY(convert_adjacency(e, g->u_Adjacent, g->u_elements, ua, "u_Adjacent"));
  TRY(convert_adjacency(e, g->v_Adjacent, g->v_elements, va, "v_Adjacent"));
  // level tables: hc*SC, CS, SC | hc*SCW, CSW, SCW (hc is REAL promoted to double)
  std::vector<double> lev(6 * LT_LEV_STRIDE, 0.0);
  const double hc = (double)e->cfg.hc;
  for (int k = 0; k < e->cfg.us; k++) {
    lev[0 * LT_LEV_STRIDE + k] = hc * g->SC[k];
    lev[1 * LT_LEV_STRIDE + k] = g->CS[k];
    lev[2 * LT_LEV_STRIDE + k] = g->SC[k];
  }
  for (int k = 0; k < e->cfg.ws; k++) {
    lev[3 * LT_LEV_STRIDE + k] = hc * g->SCW[k];
    lev[4 * LT_LEV_STRIDE + k] = g->CSW[k];
    lev[5 * LT_LEV_STRIDE + k] = g->SCW[k];
  }
  TRY(upload(e, e->rnode, rn.data(), rn.size() * sizeof(RhoNode)));
  TRY(upload(e, e->unode, un.data(), un.size() * sizeof(XYNode)));
  TRY(upload(e, e->vnode, vn.data(), vn.size() * sizeof(XYNode)));
  {
    // The internal particle order is by TILES of 32 x 32 rho elements: the waves in
    // flight on an XCD (each XCD works through a contiguous eighth of the order) then cover a few compact
    // patches whose node columns stay in its L2, where whole rows of elements (the reference's element
    // numbering) put the elements above and below a row hundreds of waves apart.  Row length from an
    // element's own corners: (i,j), (i+1,j), (i+1,j+1), (i,j+1) -> node(4) - node(1) = vi.
    const int T = getenv("LT_TILE") ? std::max(1, atoi(getenv("LT_TILE"))) : 32;   // (LT_TILE: experiments; 16 -> 1.33, 32 -> 1.30, 64 -> 1.29 GB per k_rk launch)
    // Inside a tile the elements follow a Z curve (r05): the 64 particles of a wave then cover a compact patch (at 2.4
    // particles per element 27 elements touch ~42 node columns instead of the 56 of a strip along a row), which is
    // what a sparse shard's k_rk, bound by what it fetches, is sensitive to: configs[3]'s shard 7.19 -> 7.08 ms,
    // configs[2] 4.47 -> 4.43 ms, configs[1] unchanged.  LT_TILE_ORDER=rows: rows inside the tile, as before.
    const char* tord = getenv("LT_TILE_ORDER");
    const bool morton = !(tord && !strcmp(tord, "rows")) && (T & (T - 1)) == 0 && T <= 256;
    const int32_t* RE = g->RE;
    const int vi_row = std::max(2, RE[3] - RE[0]);
    const int ntx = (vi_row + T - 1) / T;
    e->h_tile_key.resize((size_t)g->rho_elements);
    int32_t kmax = 0;
    for (int el = 0; el < g->rho_elements; el++) {
      const int n1 = RE[(size_t)el * 4] - 1, i = n1 % vi_row, j = n1 / vi_row;
      int32_t key = (((j / T) * ntx + (i / T)) * T + (j % T)) * T + (i % T);
      if (morton) {
        int32_t z = 0;
        for (int b = 0; b < 8; b++) z |= (((i % T) >> b) & 1) << (2 * b) | (((j % T) >> b) & 1) << (2 * b + 1);
        key = ((j / T) * ntx + (i / T)) * T * T + z;
      }
      e->h_tile_key[(size_t)el] = key;
      kmax = std::max(kmax, key);
    }
    e->tile_key_bits = 1;
    while (e->tile_key_bits < 31 && (1 << e->tile_key_bits) <= kmax) e->tile_key_bits++;
    TRY(upload(e, e->tile_key, e->h_tile_key.data(), e->h_tile_key.size() * 4));
  }
  TRY(upload(e, e->RE, g->RE, (size_t)g->rho_elements * 16));
  TRY(upload(e, e->UE, g->UE, (size_t)g->u_elements * 16));
  TRY(upload(e, e->VE, g->VE, (size_t)g->v_elements * 16));
  TRY(upload(e, e->rAdj, ra.data(), ra.size() * 4));
  TRY(upload(e, e->uAdj, ua.data(), ua.size() * 4));
  TRY(upload(e, e->vAdj, va.data(), va.size() * 4));
  TRY(upload(e, e->lev, lev.data(), lev.size() * 8));
  e->rho_nodes = g->rho_nodes; e->u_nodes = g->u_nodes; e->v_nodes = g->v_nodes;
  e->rho_elements = g->rho_elements; e->u_elements = g->u_elements; e->v_elements = g->v_elements;
  e->h_nx[0].assign(g->rx, g->rx + g->rho_nodes); e->h_ny[0].assign(g->ry, g->ry + g->rho_nodes);
  e->h_nx[1].assign(g->ux, g->ux + g->u_nodes);   e->h_ny[1].assign(g->uy, g->uy + g->u_nodes);
  e->h_nx[2].assign(g->vx, g->vx + g->v_nodes);   e->h_ny[2].assign(g->vy, g->vy + g->v_nodes);
  e->h_E[0].assign(g->RE, g->RE + (size_t)4 * g->rho_elements);
  e->h_E[1].assign(g->UE, g->UE + (size_t)4 * g->u_elements);
  e->h_E[2].assign(g->VE, g->VE + (size_t)4 * g->v_elements);
  e->h_adj[0].swap(ra); e->h_adj[1].swap(ua); e->h_adj[2].swap(va);
  // slabs: ALLOCATE + zero (hydrodynamic_module.f90:682-698)
  const size_t se = slab_elem(e);
  const size_t us = e->cfg.us, ws = e->cfg.ws;
  // (k_rk addresses the velocity slabs by 32-bit byte offsets)
  if ((size_t)std::max(std::max(e->u_nodes, e->v_nodes), e->rho_nodes) * ws * LT_SLOTS * se >= (size_t)0xfffff000u)
    return fail(e, LT_ERR_UNSUPPORTED, "lt_set_grid: a velocity slab of 4 GiB or more (shard the grid, or LT_SLAB_F32)");
  TRY(dev_alloc(e, e->U, (size_t)e->u_nodes * us * LT_SLOTS * se));
  TRY(dev_alloc(e, e->V, (size_t)e->v_nodes * us * LT_SLOTS * se));
  TRY(dev_alloc(e, e->W, (size_t)e->rho_nodes * ws * LT_SLOTS * se));
  TRY(dev_alloc(e, e->Kh, (size_t)e->rho_nodes * ws * LT_SLOTS * se));
  TRY(dev_alloc(e, e->zeta, (size_t)e->rho_nodes * 4 * se));
  if (needs_salt(e->cfg)) {
    TRY(dev_alloc(e, e->S, (size_t)e->rho_nodes * us * LT_SLOTS * se));
    HIPCHK(e, hipMemsetAsync(e->S.p, 0, e->S.bytes, e->stream));
  }
  if (e->cfg.SaltTempOn) {
    TRY(dev_alloc(e, e->T, (size_t)e->rho_nodes * us * LT_SLOTS * se));
    HIPCHK(e, hipMemsetAsync(e->T.p, 0, e->T.bytes, e->stream));
  }
  HIPCHK(e, hipMemsetAsync(e->U.p, 0, e->U.bytes, e->stream));
  HIPCHK(e, hipMemsetAsync(e->V.p, 0, e->V.bytes, e->stream));
  HIPCHK(e, hipMemsetAsync(e->W.p, 0, e->W.bytes, e->stream));
  HIPCHK(e, hipMemsetAsync(e->Kh.p, 0, e->Kh.bytes, e->stream));
  HIPCHK(e, hipMemsetAsync(e->zeta.p, 0, e->zeta.bytes, e->stream));
  e->have_masks = g->rho_mask && g->u_mask && g->v_mask;
  if (e->cfg.FreeSlip) {
    // the masks of every element's nodes, as the free-slip substitution reads them
    // (hydrodynamic_module.f90:1934-1994, 2329-2520): bit k = node k+1 is water
    if (!e->have_masks) return fail(e, LT_ERR_ARG, "lt_set_grid: FreeSlip needs rho_mask, u_mask, v_mask");
    auto codes = [](const int32_t* E, int ne, const int32_t* mask, int nmask) {
      std::vector<unsigned char> c((size_t)ne);
      for (int i = 0; i < ne; i++) {
        unsigned m = 0;
        for (int k = 0; k < 4; k++) {
          const int id = E[(size_t)i * 4 + k];
          // (v_mask indexed with a U node id beyond its end -- the reference reads past the array -- counts as water)
          const int w = (id >= 1 && id <= nmask) ? (mask[id - 1] != 0) : 1;
          m |= (unsigned)w << k;
        }
        c[(size_t)i] = (unsigned char)m;
      }
      return c;
    };
    const auto cr = codes(g->RE, g->rho_elements, g->rho_mask, g->rho_nodes);
    const auto cu = codes(g->UE, g->u_elements, g->u_mask, g->u_nodes);
    const auto cv = codes(g->VE, g->v_elements, g->v_mask, g->v_nodes);
    const auto cvu = codes(g->UE, g->u_elements, g->v_mask, g->v_nodes);
    TRY(upload(e, e->fsR, cr.data(), cr.size()));
    TRY(upload(e, e->fsU, cu.data(), cu.size()));
    TRY(upload(e, e->fsV, cv.data(), cv.size()));
    TRY(upload(e, e->fsVU, cvu.data(), cvu.size()));
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  if (e->have_masks) {  // m_r, m_u, m_v of updateHydro's mask multiply (lt_slab_prefetch, masked = 0)
    TRY(upload(e, e->rmask, g->rho_mask, (size_t)g->rho_nodes * 4));
    TRY(upload(e, e->umask, g->u_mask, (size_t)g->u_nodes * 4));
    TRY(upload(e, e->vmask, g->v_mask, (size_t)g->v_nodes * 4));
  }
  e->phys[1] = 0; e->phys[2] = 1; e->phys[3] = 2; e->spare = 3;
  e->prefetch_pending = false;
  size_t stage = std::max((size_t)e->rho_nodes * ws, (size_t)e->u_nodes * us) * 8;
  TRY(dev_alloc(e, e->staging, stage));
  HIPCHK(e, hipStreamSynchronize(e->stream));  // host vectors go out of scope
  e->have_grid = true;
  e->slab_set[1] = e->slab_set[2] = e->slab_set[3] = false;
  return LT_OK;
}

int lt_set_bounds(lt_engine* e, const lt_bounds* b) {
  if (!e || !b) return LT_ERR_ARG;
  if (b->nbounds < 0 || b->maxbound < 0 || b->maxisland < 0 || b->numislands < 0 ||
      (b->nbounds > 0 && (!b->bnd_x || !b->bnd_y || !b->land)) ||
      (b->maxbound > 0 && (!b->bx || !b->by)) ||
      (b->maxisland > 0 && (!b->hx || !b->hy || !b->hid)))
    return fail(e, LT_ERR_ARG, "lt_set_bounds: bad arguments");
  HIPCHK(e, hipSetDevice(e->device));
  std::vector<double> seg((size_t)b->nbounds * 4);
  for (int i = 0; i < b->nbounds; i++) {
    seg[4 * (size_t)i + 0] = b->bnd_x[2 * (size_t)i];
    seg[4 * (size_t)i + 1] = b->bnd_y[2 * (size_t)i];
    seg[4 * (size_t)i + 2] = b->bnd_x[2 * (size_t)i + 1];
    seg[4 * (size_t)i + 3] = b->bnd_y[2 * (size_t)i + 1];
  }
  HostIndex hs, hm, hi;
  build_seg_index(b->bnd_x, b->bnd_y, b->nbounds, e->seg, hs);
  const int32_t* hid = b->hid;
  const bool pip_ok = build_pip_index(b->bx, b->by, nullptr, b->maxbound, [](int) { return true; }, e->pip_main, hm) &&
                      build_pip_index(b->hx, b->hy, hid, b->maxisland, [hid](int i) { return hid[i] == hid[i + 1]; }, e->pip_isl, hi);
  if (!pip_ok) e->use_pip_index = 0;   // (polygons whose slab lists would not fit: the reference's full algorithm)
  TRY(upload(e, e->bnd, seg.data(), seg.size() * 8));
  TRY(upload(e, e->land, b->land, (size_t)b->nbounds * 4));
  TRY(upload(e, e->bx, b->bx, (size_t)b->maxbound * 8));
  TRY(upload(e, e->by, b->by, (size_t)b->maxbound * 8));
  TRY(upload(e, e->hx, b->hx, (size_t)b->maxisland * 8));
  TRY(upload(e, e->hy, b->hy, (size_t)b->maxisland * 8));
  TRY(upload(e, e->hid, b->hid, (size_t)b->maxisland * 4));
  TRY(upload(e, e->seg_ptr, hs.ptr.data(), hs.ptr.size() * 4));
  TRY(upload(e, e->seg_ids, hs.ids.data(), hs.ids.size() * 4));
  TRY(upload(e, e->pm_ptr, hm.ptr.data(), hm.ptr.size() * 4));
  TRY(upload(e, e->pm_ids, hm.gidx.data(), hm.gidx.size() * 4));
  TRY(upload(e, e->pi_ptr, hi.ptr.data(), hi.ptr.size() * 4));
  TRY(upload(e, e->pi_ids, hi.gidx.data(), hi.gidx.size() * 4));
  TRY(upload(e, e->pm_rec, hm.rec.data(), hm.rec.size() * 8));
  TRY(upload(e, e->pi_rec, hi.rec.data(), hi.rec.size() * 8));
  TRY(upload(e, e->pi_pid, hi.pid.data(), hi.pid.size() * 4));
  TRY(upload(e, e->pm_ys, hm.ys.data(), hm.ys.size() * 8));
  TRY(upload(e, e->pi_ys, hi.ys.data(), hi.ys.size() * 8));
  e->seg.ptr = (const int*)e->seg_ptr.p; e->seg.ids = (const int*)e->seg_ids.p;
  e->pip_main.ptr = (const int*)e->pm_ptr.p; e->pip_main.gidx = (const int*)e->pm_ids.p;
  e->pip_isl.ptr = (const int*)e->pi_ptr.p; e->pip_isl.gidx = (const int*)e->pi_ids.p;
  e->pip_main.rec = (const double4*)e->pm_rec.p; e->pip_main.pid = nullptr; e->pip_main.ys = (const double*)e->pm_ys.p;
  e->pip_isl.rec = (const double4*)e->pi_rec.p; e->pip_isl.pid = (const int*)e->pi_pid.p; e->pip_isl.ys = (const double*)e->pi_ys.p;
  e->nbounds = b->nbounds; e->maxbound = b->maxbound; e->maxisland = b->maxisland;
  e->numislands = b->numislands;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->have_bounds = true;
  return LT_OK;
}

int lt_set_habitat(lt_engine* e, const lt_habitat* h) {
  if (!e || !h) return LT_ERR_ARG;
  if (!e->have_grid) return fail(e, LT_ERR_STATE, "lt_set_habitat before lt_set_grid");
  if (h->pedges <= 0 || !h->polys || !h->polyspecs || !h->maxbdis || !h->elepolys_ptr ||
      !h->polyholes_ptr || h->maxpolyid < h->minpolyid)
    return fail(e, LT_ERR_ARG, "lt_set_habitat: bad arguments");
  HIPCHK(e, hipSetDevice(e->device));
  e->hab = *h;
  e->npoly = h->maxpolyid - h->minpolyid + 1;
  e->nhole = (h->hedges > 0) ? h->maxholeid - h->minholeid + 1 : 0;
  int nlist = h->elepolys_ptr[e->rho_elements], nph = h->polyholes_ptr[e->npoly];
  TRY(upload(e, e->polys, h->polys, (size_t)h->pedges * 5 * 8));
  TRY(upload(e, e->holes, h->holes, (size_t)h->hedges * 6 * 8));
  TRY(upload(e, e->polyspecs, h->polyspecs, (size_t)e->npoly * 2 * 4));
  TRY(upload(e, e->holespecs, h->holespecs, (size_t)e->nhole * 2 * 4));
  TRY(upload(e, e->maxbdis, h->maxbdis, (size_t)e->npoly * 8));
  TRY(upload(e, e->maxhdis, h->maxhdis, (size_t)e->nhole * 8));
  TRY(upload(e, e->ep_ptr, h->elepolys_ptr, ((size_t)e->rho_elements + 1) * 4));
  TRY(upload(e, e->ep_ids, h->elepolys_ids, (size_t)nlist * 4));
  TRY(upload(e, e->ph_ptr, h->polyholes_ptr, ((size_t)e->npoly + 1) * 4));
  TRY(upload(e, e->ph_ids, h->polyholes_ids, (size_t)nph * 4));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->h_polyspecs.assign(h->polyspecs, h->polyspecs + (size_t)e->npoly * 2);
  if (e->nhole > 0 && h->holespecs) e->h_holespecs.assign(h->holespecs, h->holespecs + (size_t)e->nhole * 2);
  else e->h_holespecs.clear();
  e->h_ep_ptr.assign(h->elepolys_ptr, h->elepolys_ptr + (size_t)e->rho_elements + 1);
  e->h_ep_ids.assign(h->elepolys_ids, h->elepolys_ids + nlist);
  e->h_ph_ptr.assign(h->polyholes_ptr, h->polyholes_ptr + (size_t)e->npoly + 1);
  e->h_ph_ids.assign(h->polyholes_ids, h->polyholes_ids + nph);
  e->have_habitat = true;
  return LT_OK;
}

// ---- init-time geometry on the device (lt_init.h) ---------------------------------------
int lt_build_adjacency(int32_t device, int32_t elements, int32_t row, const int32_t* E, int32_t* adjacent) {
  if (elements <= 0 || row < 0 || !E || !adjacent) return LT_ERR_ARG;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return LT_ERR_DEVICE;
  int4* dE = nullptr;
  int *dA = nullptr, *dO = nullptr;
  const size_t be = (size_t)elements * 16, ba = (size_t)elements * 10 * 4;
  int rc = LT_OK, overflow = 0;
  if (hipMalloc(&dE, be) != hipSuccess || hipMalloc(&dA, ba) != hipSuccess || hipMalloc(&dO, 4) != hipSuccess) {
    rc = LT_ERR_NOMEM;
  } else if (hipMemcpy(dE, E, be, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dA, 0, ba) != hipSuccess ||
             hipMemset(dO, 0, 4) != hipSuccess) {
    rc = LT_ERR_DEVICE;
  } else {
    hipLaunchKernelGGL(ltd::k_adjacency, dim3((unsigned)((elements + 255) / 256)), dim3(256), 0, 0, dE, (int)elements,
                       (int)row, dA, dO);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(adjacent, dA, ba, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&overflow, dO, 4, hipMemcpyDeviceToHost) != hipSuccess)
      rc = LT_ERR_DEVICE;
    else if (overflow)
      rc = LT_ERR_ARG;  // an element with more than 9 node-sharing neighbours in the window
  }
  (void)hipFree(dE); (void)hipFree(dA); (void)hipFree(dO);
  return rc;
}

int lt_locate_particles(lt_engine* e, int32_t n, const double* Xpar, const double* Ypar, int32_t* P_r_element,
                        int32_t* P_u_element, int32_t* P_v_element, int32_t* err, int32_t* par) {
  if (!e) return LT_ERR_ARG;
  if (n < 0 || (n > 0 && (!Xpar || !Ypar))) return fail(e, LT_ERR_ARG, "lt_locate_particles: bad arguments");
  if (!e->have_grid) return fail(e, LT_ERR_STATE, "lt_locate_particles before lt_set_grid");
  HIPCHK(e, hipSetDevice(e->device));
  int32_t* outp[3] = {P_r_element, P_u_element, P_v_element};
  const int nele[3] = {e->rho_elements, e->u_elements, e->v_elements};
  const void* nodes[3] = {e->rnode.p, e->unode.p, e->vnode.p};
  const int4* EL[3] = {(const int4*)e->RE.p, (const int4*)e->UE.p, (const int4*)e->VE.p};
  std::vector<int> codes[3], cnt[3];
  if (n > 0) {
    DevBuf dX, dY, dcodes, dcnt, dptr, dids;
    int rc = LT_OK;
    auto body = [&]() -> int {
      TRY(upload(e, dX, Xpar, (size_t)n * 8));
      TRY(upload(e, dY, Ypar, (size_t)n * 8));
      TRY(dev_alloc(e, dcodes, (size_t)n * LT_LOC_CAP * 4));
      TRY(dev_alloc(e, dcnt, (size_t)n * 4));
      for (int g = 0; g < 3; g++) {
        ltd::EleBins B;
        HostIndex H;
        ltih::BinGeom G;
        ltih::build_ele_bins(e->h_nx[g], e->h_ny[g], e->h_E[g], nele[g], G, H.ptr, H.ids);
        B.x0 = G.x0; B.y0 = G.y0; B.inv = G.inv; B.ncx = G.ncx; B.ncy = G.ncy;
        if (H.ids.empty()) H.ids.push_back(0);
        TRY(upload(e, dptr, H.ptr.data(), H.ptr.size() * 4));
        TRY(upload(e, dids, H.ids.data(), H.ids.size() * 4));
        B.ptr = (const int*)dptr.p; B.ids = (const int*)dids.p;
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (g == 0)
          hipLaunchKernelGGL(ltd::k_locate<true>, grid, block, 0, e->stream, nodes[g], EL[g], B, (const double*)dX.p,
                             (const double*)dY.p, (int)n, (int*)dcodes.p, (int*)dcnt.p);
        else
          hipLaunchKernelGGL(ltd::k_locate<false>, grid, block, 0, e->stream, nodes[g], EL[g], B, (const double*)dX.p,
                             (const double*)dY.p, (int)n, (int*)dcodes.p, (int*)dcnt.p);
        HIPCHK(e, hipGetLastError());
        codes[g].resize((size_t)n * LT_LOC_CAP);
        cnt[g].resize((size_t)n);
        HIPCHK(e, hipMemcpyAsync(codes[g].data(), dcodes.p, (size_t)n * LT_LOC_CAP * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(cnt[g].data(), dcnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
      }
      return LT_OK;
    };
    rc = body();
    for (DevBuf* b : {&dX, &dY, &dcodes, &dcnt, &dptr, &dids}) if (b->p) (void)hipFree(b->p);
    if (rc != LT_OK) return rc;
    for (int g = 0; g < 3; g++)
      for (int k = 0; k < n; k++)
        if (cnt[g][k] > LT_LOC_CAP)
          return fail(e, LT_ERR_ARG, "lt_locate_particles: particle %d lies in more than %d elements", k + 1, LT_LOC_CAP);
  }
  // the particle loop of setEle_all: only the ORDER of the tests depends on the previous particle (lt_init_host.h)
  std::vector<int32_t> out[3];
  int error = 0, p = 0;
  ltih::replay_set_ele_all((int)n, codes, cnt, e->h_adj, out, error, p);
  for (int g = 0; g < 3; g++)
    if (outp[g] && n > 0) memcpy(outp[g], out[g].data(), (size_t)n * 4);
  if (err) *err = error;
  if (par) *par = p;
  return LT_OK;
}

int lt_create_polyspecs(lt_engine* e, const lt_habitat* h) {
  if (!e || !h) return LT_ERR_ARG;
  if (!e->have_grid) return fail(e, LT_ERR_STATE, "lt_create_polyspecs before lt_set_grid");
  if (h->pedges <= 0 || !h->polys || !h->maxbdis || h->maxpolyid < h->minpolyid ||
      (h->hedges > 0 && (!h->holes || h->maxholeid < h->minholeid)))
    return fail(e, LT_ERR_ARG, "lt_create_polyspecs: bad arguments");
  HIPCHK(e, hipSetDevice(e->device));
  const int pedges = h->pedges, hedges = h->hedges;
  const int npoly = h->maxpolyid - h->minpolyid + 1;
  const int nhole = (hedges > 0) ? h->maxholeid - h->minholeid + 1 : 0;
  // polyspecs / holespecs and the holes of every polygon: serial scans (lt_init_host.h)
  std::vector<int32_t> ps, hs, php, phi;
  {
    long bad = 0;
    const int rc = ltih::polyspecs_host(h->polys, pedges, h->minpolyid, h->maxpolyid, h->holes, hedges, h->minholeid,
                                        h->maxholeid, ps, hs, php, phi, bad);
    if (rc == 1) return fail(e, LT_ERR_ARG, "lt_create_polyspecs: polygon id %ld outside minpolyid..maxpolyid", bad);
    if (rc == 2) return fail(e, LT_ERR_ARG, "lt_create_polyspecs: hole id %ld outside minholeid..maxholeid", bad);
  }
  // polygons of every rho element :293-386, on the device
  TRY(upload(e, e->polys, h->polys, (size_t)pedges * 5 * 8));
  TRY(upload(e, e->holes, h->holes, (size_t)hedges * 6 * 8));
  TRY(upload(e, e->polyspecs, ps.data(), (size_t)npoly * 2 * 4));
  TRY(upload(e, e->holespecs, hs.data(), (size_t)nhole * 2 * 4));
  TRY(upload(e, e->maxbdis, h->maxbdis, (size_t)npoly * 8));
  TRY(upload(e, e->maxhdis, h->maxhdis, (size_t)(h->maxhdis ? nhole : 0) * 8));
  const int ne = e->rho_elements;
  std::vector<int32_t> epp((size_t)ne + 1, 0), epi;
  {
    DevBuf dcount;
    TRY(dev_alloc(e, dcount, (size_t)ne * 4));
    const dim3 grid((unsigned)((ne + 255) / 256)), block(256);
    hipLaunchKernelGGL(ltd::k_elepolys, grid, block, 0, e->stream, (const RhoNode*)e->rnode.p, (const int4*)e->RE.p, ne,
                       (const double*)e->polys.p, pedges, (const int*)e->polyspecs.p, (const double*)e->maxbdis.p,
                       (int)h->minpolyid, npoly, (const int*)nullptr, (int*)dcount.p, 0);
    std::vector<int32_t> counts((size_t)ne);
    hipError_t he = hipMemcpyAsync(counts.data(), dcount.p, (size_t)ne * 4, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void)hipFree(dcount.p);
    if (he != hipSuccess) return fail(e, LT_ERR_DEVICE, "lt_create_polyspecs: %s", hipGetErrorString(he));
    for (int i = 0; i < ne; i++) epp[(size_t)i + 1] = epp[(size_t)i] + counts[(size_t)i];
    epi.assign((size_t)std::max(epp[(size_t)ne], 1), 0);
    TRY(upload(e, e->ep_ptr, epp.data(), epp.size() * 4));
    TRY(dev_alloc(e, e->ep_ids, epi.size() * 4));
    hipLaunchKernelGGL(ltd::k_elepolys, grid, block, 0, e->stream, (const RhoNode*)e->rnode.p, (const int4*)e->RE.p, ne,
                       (const double*)e->polys.p, pedges, (const int*)e->polyspecs.p, (const double*)e->maxbdis.p,
                       (int)h->minpolyid, npoly, (const int*)e->ep_ptr.p, (int*)e->ep_ids.p, 1);
    HIPCHK(e, hipMemcpyAsync(epi.data(), e->ep_ids.p, epi.size() * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    epi.resize((size_t)epp[(size_t)ne]);
  }
  TRY(upload(e, e->ph_ptr, php.data(), php.size() * 4));
  TRY(upload(e, e->ph_ids, phi.data(), phi.size() * 4));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->hab = *h;
  e->hab.polyspecs = e->hab.holespecs = e->hab.elepolys_ptr = e->hab.elepolys_ids = nullptr;
  e->hab.polyholes_ptr = e->hab.polyholes_ids = nullptr;
  e->npoly = npoly;
  e->nhole = nhole;
  e->h_polyspecs.swap(ps);
  if (nhole > 0) e->h_holespecs.swap(hs); else e->h_holespecs.clear();
  e->h_ep_ptr.swap(epp); e->h_ep_ids.swap(epi); e->h_ph_ptr.swap(php); e->h_ph_ids.swap(phi);
  e->have_habitat = true;
  return LT_OK;
}

int lt_get_polyspecs(lt_engine* e, int32_t* polyspecs, int32_t* holespecs, int32_t* elepolys_ptr, int32_t* elepolys_ids,
                     int32_t elepolys_cap, int32_t* polyholes_ptr, int32_t* polyholes_ids, int32_t polyholes_cap) {
  if (!e) return LT_ERR_ARG;
  if (!e->have_habitat) return fail(e, LT_ERR_STATE, "lt_get_polyspecs: no habitat");
  auto put = [](int32_t* dst, const std::vector<int32_t>& v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * 4); };
  put(polyspecs, e->h_polyspecs);
  put(holespecs, e->h_holespecs);
  put(elepolys_ptr, e->h_ep_ptr);
  put(polyholes_ptr, e->h_ph_ptr);
  if ((elepolys_ids && (size_t)std::max(elepolys_cap, 0) < e->h_ep_ids.size()) ||
      (polyholes_ids && (size_t)std::max(polyholes_cap, 0) < e->h_ph_ids.size()))
    return fail(e, LT_ERR_ARG, "lt_get_polyspecs: id arrays too small (%zu, %zu needed)", e->h_ep_ids.size(), e->h_ph_ids.size());
  put(elepolys_ids, e->h_ep_ids);
  put(polyholes_ids, e->h_ph_ids);
  return LT_OK;
}

int lt_set_particles(lt_engine* e, const lt_particles* p) {
  if (!e || !p) return LT_ERR_ARG;
  if (e->pre_valid) return fail(e, LT_ERR_STATE, "not between lt_count_draws and its lt_step");
  if (!e->have_grid) return fail(e, LT_ERR_STATE, "lt_set_particles before lt_set_grid");
  if (!e->par.p && !p->par) return fail(e, LT_ERR_ARG, "lt_set_particles: par is required");
  HIPCHK(e, hipSetDevice(e->device));
  e->rng_static_own = -1;   // (new particle state: a static stream window reads the shard's draws back again)
  const size_t n = (size_t)e->cfg.numpar;
  if (p->P_r_element) TRY(check_particle_elements(e, p->P_r_element, (int)n, e->rho_elements, "P_r_element"));
  if (p->P_u_element) TRY(check_particle_elements(e, p->P_u_element, (int)n, e->u_elements, "P_u_element"));
  if (p->P_v_element) TRY(check_particle_elements(e, p->P_v_element, (int)n, e->v_elements, "P_v_element"));
  if (!e->pr.p && !(p->P_r_element && p->P_u_element && p->P_v_element))
    return fail(e, LT_ERR_ARG, "lt_set_particles: element ids are required on first upload");
  if (e->perm.size() != n) {
    // first upload: choose the internal order.  Particles of one rho element are
    // made neighbours (stable, so equal keys keep the reference order): the lanes
    // of a wave then gather from the same few cache lines of the slabs.
    e->perm.resize(n);
    for (size_t i = 0; i < n; i++) e->perm[i] = (int)i;
    if (e->sort_particles && p->P_r_element) {
      const int32_t* el = p->P_r_element;
      const int32_t* tk = e->h_tile_key.data();
      std::stable_sort(e->perm.begin(), e->perm.end(), [el, tk](int a, int b) { return tk[el[a] - 1] < tk[el[b] - 1]; });
    }
    TRY(upload(e, e->orig, e->perm.data(), n * 4));
    TRY(dev_alloc(e, e->ws_d, (size_t)WS_NDOUBLE * n * 8));
    TRY(dev_alloc(e, e->ws_i, n * 4));
    TRY(dev_alloc(e, e->slow_list, (size_t)n * 4));
    if (e->cfg.HTurbOn || e->cfg.VTurbOn || (e->cfg.Behavior >= 1 && e->cfg.Behavior <= 5)) {
      TRY(dev_alloc(e, e->flag, (n + 1) * 4));
      TRY(dev_alloc(e, e->rank, (n + 1) * 4));
    }
    if (e->cfg.VTurbOn) {
      {
        // VTurb's working set per particle: 4*ws knots x (24-byte record + 8 bytes for a tension factor + 26 bytes of SIGS
        // task slots) + the idt/2 (+2) deviates + 12 bytes of walk list + 32 bytes of flags -- 7.7 KB at ws = 31, 12.3 KB at ws = 51.  Said
        // here, with the figures, rather than by whichever hipMalloc happens to fail.
        const size_t per = (size_t)4 * e->cfg.ws * (32 + 26) + (size_t)(std::max(e->cfg.idt / 2, 1) + VT_DEV_HROWS) * 8 + 12 + 32;
        const size_t want = per * ((n + 63) / 64 * 64);
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && want > mfree)
          return fail(e, LT_ERR_NOMEM, "lt_set_particles: vertical turbulence needs %.1f GB for %zu particles at ws = %d "
                      "(%zu bytes each: spline table, SIGS task lists, deviates) and %.1f GB are free on device %d: shard the "
                      "particles over more engines", want / 1e9, n, (int)e->cfg.ws, per, mfree / 1e9, e->device);
      }
      // p2 = 4*ws knots per particle, [wave][knot][lane] records {x, y, yp}
      TRY(dev_alloc(e, e->vt_tab, (size_t)4 * e->cfg.ws * ((n + 63) / 64 * 64) * (sizeof(VtRec) + 8)));   // {x, y, yp} | tension factors
      TRY(dev_alloc(e, e->vt_bits, ((n + 63) / 64) * 4 * 64 * 8));
      // task lists of the SIGS iterations
      const size_t waves = (n + 63) / 64;
      // (room for every interval of every lane: 4*ws - 1 per profile.  ~9 % of them are listed on the bench
      // workload, but a list that cannot overflow lets k_vt_table do without an in-lane SIGS iteration -- a real
      // function call in its knot loop, with the scalar registers and the stack slots that go with one)
      e->vt_qcap = 64 * 4 * e->cfg.ws;
      TRY(dev_alloc(e, e->vt_qS, waves * e->vt_qcap * 8)); TRY(dev_alloc(e, e->vt_qS1, waves * e->vt_qcap * 8));
      TRY(dev_alloc(e, e->vt_qS2, waves * e->vt_qcap * 8)); TRY(dev_alloc(e, e->vt_qwl, waves * e->vt_qcap * 2));
      TRY(dev_alloc(e, e->vt_qcount, waves * 4 + 4));
      TRY(dev_alloc(e, e->vt_wl, (3 * n + 4) * 4));   // count | list (a stop per particle + a SigErr entry per particle) | resume
      TRY(dev_alloc(e, e->vt_dev, waves * 64 * (size_t)(std::max(e->cfg.idt / 2, 1) + VT_DEV_HROWS) * 8));
    }
  }
  TRY(sync_perm(e));
  // positions / elements may change: nothing is known to be verified any more
  TRY(dev_alloc(e, e->ele_ok, n));
  TRY(dev_alloc(e, e->block_active, ((size_t)n + 255) / 256 * 4 + 4));
  HIPCHK(e, hipMemsetAsync(e->ele_ok.p, 0, e->ele_ok.bytes, e->stream));
  TRY(put_field<double>(e, e->par, p->par, 13));
  TRY(put_field<int32_t>(e, e->pr, p->P_r_element, 1));
  TRY(put_field<int32_t>(e, e->pu, p->P_u_element, 1));
  TRY(put_field<int32_t>(e, e->pv, p->P_v_element, 1));
  TRY(put_field<int32_t>(e, e->P_behave, p->P_behave, 1));
  TRY(put_field<int32_t>(e, e->dead, p->dead, 1));
  TRY(put_field<int32_t>(e, e->oob, p->oob, 1));
  TRY(put_field<double>(e, e->P_swim, p->P_swim, 3));
  TRY(put_field<double>(e, e->P_pediage, p->P_pediage, 1));
  TRY(put_field<double>(e, e->P_deadage, p->P_deadage, 1));
  TRY(put_field<int32_t>(e, e->settle, p->settle, 1));
  TRY(put_field<double>(e, e->settletime, p->settletime, 1));
  TRY(put_field<int32_t>(e, e->endpoly, p->endpoly, 1));
  TRY(put_field<int32_t>(e, e->hitBottom, p->hitBottom, 1));
  TRY(put_field<int32_t>(e, e->hitLand, p->hitLand, 1));
  TRY(put_field<int32_t>(e, e->err_code, p->err_code, 1));
  TRY(put_field<double>(e, e->timer, p->timer, 1));
  TRY(put_field<double>(e, e->P_Sprev, p->P_Sprev, 1));
  TRY(put_field<double>(e, e->P_zprev, p->P_zprev, 1));
  {
    const bool fresh = !e->bottom.p;
    TRY(put_field<int32_t>(e, e->bottom, p->bottom, 1));
    if (fresh && !p->bottom && e->cfg.Behavior == 7) {  // bottom = .TRUE., behavior_module.f90:113
      std::vector<int32_t> ones((size_t)n, 1);
      // (on the engine's stream, behind put_field's memset: a null-stream copy is not ordered against it)
      HIPCHK(e, hipMemcpyAsync(e->bottom.p, ones.data(), (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
    }
  }
  TRY(put_field<double>(e, e->P_Salt, p->P_Salt, 1));
  TRY(put_field<double>(e, e->P_Temp, p->P_Temp, 1));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->have_particles = true;
  return LT_OK;
}

static int fetch_counters(lt_engine* e);
// Under a static stream window (lt_set_rng_window_static) the calls that hand positions out without waiting for the
// device -- lt_pack_positions, lt_allgather_positions -- look at the word k_post counts broken windows in without
// waiting either: an asynchronous copy per call, examined by the next one (the report is one call late, not absent).
static int check_static_miss_async(lt_engine* e) {
  if (!e->rng_static) return LT_OK;
  if (!e->h_miss) {
    // (the event first: h_miss is only published together with an event that can be recorded)
    if (!e->miss_ev && hipEventCreateWithFlags(&e->miss_ev, hipEventDisableTiming) != hipSuccess) {
      e->miss_ev = nullptr;
      return fail(e, LT_ERR_DEVICE, "hipEventCreate failed (static-window check)");
    }
    unsigned long long* hm = nullptr;
    if (hipHostMalloc((void**)&hm, sizeof(unsigned long long)) != hipSuccess) return fail(e, LT_ERR_NOMEM, "pinned word (static-window check)");
    *hm = 0ull;
    e->h_miss = hm;
  }
  if (e->miss_inflight && hipEventQuery(e->miss_ev) == hipSuccess) {
    e->miss_inflight = false;
    if (*e->h_miss) {
      // said here: the lt_set_rng_window that ends the window does not say it a second time
      if (e->h_counters) e->h_counters[CNT_STATIC_MISS] = *e->h_miss;
      e->miss_reported = true;
    }
    if (*e->h_miss)
      return fail(e, LT_ERR_STATE, "static random-stream window: the shard's draws changed (%llu step(s)); a particle "
                                   "stopped or started drawing -- set the window before each step instead", *e->h_miss);
  }
  if (!e->miss_inflight) {
    HIPCHK(e, hipMemcpyAsync(e->h_miss, (unsigned long long*)e->counters.p + CNT_STATIC_MISS, sizeof(unsigned long long),
                             hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipEventRecord(e->miss_ev, e->stream));
    e->miss_inflight = true;
  }
  return LT_OK;
}
int lt_get_particles(lt_engine* e, lt_particles* out) {
  if (!e || !out) return LT_ERR_ARG;
  if (!e->have_particles) return fail(e, LT_ERR_STATE, "no particles uploaded");
  HIPCHK(e, hipSetDevice(e->device));
  if (e->rng_static) TRY(fetch_counters(e));  // (reports a static stream window that did not hold)
  {
    // the staging buffer: one part per array that is asked for and exists
    const size_t n = (size_t)e->cfg.numpar;
    size_t total = 0;
    auto want = [&](const DevBuf& b, const void* dst, size_t elem, int cols) {
      if (dst && b.p) total += ((n * cols * elem) + 255) & ~(size_t)255;
    };
    want(e->par, out->par, 8, 13); want(e->pr, out->P_r_element, 4, 1); want(e->pu, out->P_u_element, 4, 1);
    want(e->pv, out->P_v_element, 4, 1); want(e->P_behave, out->P_behave, 4, 1); want(e->dead, out->dead, 4, 1);
    want(e->oob, out->oob, 4, 1); want(e->P_swim, out->P_swim, 8, 3); want(e->P_pediage, out->P_pediage, 8, 1);
    want(e->P_deadage, out->P_deadage, 8, 1); want(e->settle, out->settle, 4, 1); want(e->settletime, out->settletime, 8, 1);
    want(e->endpoly, out->endpoly, 4, 1); want(e->hitBottom, out->hitBottom, 4, 1); want(e->hitLand, out->hitLand, 4, 1);
    want(e->err_code, out->err_code, 4, 1); want(e->timer, out->timer, 8, 1); want(e->P_Sprev, out->P_Sprev, 8, 1);
    want(e->P_zprev, out->P_zprev, 8, 1); want(e->bottom, out->bottom, 4, 1); want(e->P_Salt, out->P_Salt, 8, 1);
    want(e->P_Temp, out->P_Temp, 8, 1);
    TRY(get_fields_begin(e, total));
  }
  TRY(get_field<double>(e, e->par, out->par, 13));
  TRY(get_field<int32_t>(e, e->pr, out->P_r_element, 1));
  TRY(get_field<int32_t>(e, e->pu, out->P_u_element, 1));
  TRY(get_field<int32_t>(e, e->pv, out->P_v_element, 1));
  TRY(get_field<int32_t>(e, e->P_behave, out->P_behave, 1));
  TRY(get_field<int32_t>(e, e->dead, out->dead, 1));
  TRY(get_field<int32_t>(e, e->oob, out->oob, 1));
  TRY(get_field<double>(e, e->P_swim, out->P_swim, 3));
  TRY(get_field<double>(e, e->P_pediage, out->P_pediage, 1));
  TRY(get_field<double>(e, e->P_deadage, out->P_deadage, 1));
  TRY(get_field<int32_t>(e, e->settle, out->settle, 1));
  TRY(get_field<double>(e, e->settletime, out->settletime, 1));
  TRY(get_field<int32_t>(e, e->endpoly, out->endpoly, 1));
  TRY(get_field<int32_t>(e, e->hitBottom, out->hitBottom, 1));
  TRY(get_field<int32_t>(e, e->hitLand, out->hitLand, 1));
  TRY(get_field<int32_t>(e, e->err_code, out->err_code, 1));
  TRY(get_field<double>(e, e->timer, out->timer, 1));
  TRY(get_field<double>(e, e->P_Sprev, out->P_Sprev, 1));
  TRY(get_field<double>(e, e->P_zprev, out->P_zprev, 1));
  TRY(get_field<int32_t>(e, e->bottom, out->bottom, 1));
  TRY(get_field<double>(e, e->P_Salt, out->P_Salt, 1));
  TRY(get_field<double>(e, e->P_Temp, out->P_Temp, 1));
  return get_fields_end(e);
}

int lt_get_output(lt_engine* e, lt_output* out, int32_t reset_hits) {
  if (!e || !out) return LT_ERR_ARG;
  if (!e->have_particles) return fail(e, LT_ERR_STATE, "no particles uploaded");
  if (e->pre_valid) return fail(e, LT_ERR_STATE, "not between lt_count_draws and its lt_step");
  HIPCHK(e, hipSetDevice(e->device));
  if (e->rng_static) TRY(fetch_counters(e));  // (reports a static stream window that did not hold; the call waits anyway)
  const size_t n = (size_t)e->cfg.numpar;
  if (!n) return LT_OK;
  TRY(dev_alloc(e, e->out_cols, 5 * n * 8));
  TRY(dev_alloc(e, e->out_hits, 2 * n * 4));
  StepParams P;
  fill_params(e, P);
  hipLaunchKernelGGL(k_output, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, P, e->cfg.Earth_Radius,
                     e->cfg.SphericalProjection, e->cfg.lonmin, e->cfg.latmin, (double*)e->out_cols.p,
                     (int*)e->out_hits.p, (int)reset_hits);
  HIPCHK(e, hipGetLastError());
  double* dst[5] = {out->lon, out->lat, out->z, out->age, out->status};
  for (int c = 0; c < 5; c++)
    if (dst[c]) HIPCHK(e, hipMemcpyAsync(dst[c], (double*)e->out_cols.p + c * n, n * 8, hipMemcpyDeviceToHost, e->stream));
  if (out->hitBottom) HIPCHK(e, hipMemcpyAsync(out->hitBottom, (int*)e->out_hits.p, n * 4, hipMemcpyDeviceToHost, e->stream));
  if (out->hitLand) HIPCHK(e, hipMemcpyAsync(out->hitLand, (int*)e->out_hits.p + n, n * 4, hipMemcpyDeviceToHost, e->stream));
  if (out->salt || out->temp) {
    TRY(get_fields_begin(e, 2 * ((n * 8 + 255) & ~(size_t)255)));
    if (out->salt) TRY(get_field<double>(e, e->P_Salt, out->salt, 1));
    if (out->temp) TRY(get_field<double>(e, e->P_Temp, out->temp, 1));
    return get_fields_end(e);
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return LT_OK;
}

int lt_set_slab(lt_engine* e, int32_t slot, const lt_slab* s) {
  if (!e || !s) return LT_ERR_ARG;
  if (e->pre_valid) return fail(e, LT_ERR_STATE, "not between lt_count_draws and its lt_step");
  if (!e->have_grid) return fail(e, LT_ERR_STATE, "lt_set_slab before lt_set_grid");
  if (slot < 1 || slot > 3 || !s->zeta || !s->Uvel || !s->Vvel || !s->Wvel ||
      (e->cfg.VTurbOn && !s->Kh) || (needs_salt(e->cfg) && !s->salt) || (e->cfg.SaltTempOn && !s->temp))
    return fail(e, LT_ERR_ARG, "lt_set_slab: bad arguments (Kh is required when VTurbOn, salt by behaviours "
                               "4,5,7 and SaltTempOn, temp by SaltTempOn)");
  HIPCHK(e, hipSetDevice(e->device));
  int rc = (e->cfg.slab_storage == LT_SLAB_F32) ? set_slab_t<float>(e, slot, s)
                                                : set_slab_t<double>(e, slot, s);
  if (rc == LT_OK) e->slab_set[slot] = true;
  return rc;
}

int lt_slab_prefetch(lt_engine* e, const lt_rom_record* r) {
  if (!e || !r) return LT_ERR_ARG;
  if (r->struct_size != (int32_t)sizeof(lt_rom_record)) return fail(e, LT_ERR_ABI, "lt_rom_record size mismatch");
  if (!e->have_grid) return fail(e, LT_ERR_STATE, "lt_slab_prefetch before lt_set_grid");
  if (e->prefetch_pending) return fail(e, LT_ERR_STATE, "lt_slab_prefetch: the previous record has not been committed");
  if ((r->dtype != LT_ROM_F64 && r->dtype != LT_ROM_F32) || !r->romZ || !r->romU || !r->romV || !r->romW ||
      (e->cfg.VTurbOn && !r->romKH) || (needs_salt(e->cfg) && !r->romS) || (e->cfg.SaltTempOn && !r->romT))
    return fail(e, LT_ERR_ARG, "lt_slab_prefetch: bad arguments (romKH is required when VTurbOn, romS by "
                               "behaviours 4,5,7 and SaltTempOn, romT by SaltTempOn)");
  if ((long long)r->vi * r->uj != e->rho_nodes || (long long)r->ui * r->uj != e->u_nodes ||
      (long long)r->vi * r->vj != e->v_nodes || r->vi < 1 || r->uj < 1 || r->ui < 1 || r->vj < 1)
    return fail(e, LT_ERR_ARG, "lt_slab_prefetch: vi,uj,ui,vj do not match the node counts of lt_set_grid");
  if (!r->masked && !e->have_masks)
    return fail(e, LT_ERR_ARG, "lt_slab_prefetch: masked = 0 needs rho_mask/u_mask/v_mask in lt_set_grid");
  const int us = e->cfg.us, ws = e->cfg.ws;
  int w[12] = {1, r->vi, 1, r->uj, 1, r->ui, 1, r->uj, 1, r->vi, 1, r->vj};
  if (r->has_window) {
    const int hi[12] = {r->vi, r->vi, r->uj, r->uj, r->ui, r->ui, r->uj, r->uj, r->vi, r->vi, r->vj, r->vj};
    for (int i = 0; i < 12; i++) {
      w[i] = r->window[i];
      if (w[i] < 1 || w[i] > hi[i] || ((i & 1) && w[i] < w[i - 1]))
        return fail(e, LT_ERR_ARG, "lt_slab_prefetch: window(%d) = %d is outside the grid", i + 1, w[i]);
    }
  }
  const int* mr = r->masked ? nullptr : (const int*)e->rmask.p;
  const int* mu = r->masked ? nullptr : (const int*)e->umask.p;
  const int* mv = r->masked ? nullptr : (const int*)e->vmask.p;
  RomField f[7] = {
      {r->romZ, mr, &e->zeta, r->vi, r->uj, 1, w[0], w[1], w[2], w[3], 0, 0},
      {r->romU, mu, &e->U, r->ui, r->uj, us, w[4], w[5], w[6], w[7], 0, LT_LAYOUT_LEVFAST},
      {r->romV, mv, &e->V, r->vi, r->vj, us, w[8], w[9], w[10], w[11], 0, LT_LAYOUT_LEVFAST},
      {r->romW, mr, &e->W, r->vi, r->uj, ws, w[0], w[1], w[2], w[3], 0, LT_LAYOUT_LEVFAST}};
  int nf = 4;
  if (r->romKH) f[nf++] = {r->romKH, mr, &e->Kh, r->vi, r->uj, ws, w[0], w[1], w[2], w[3], 0, 0};
  if (r->romS && e->S.p) f[nf++] = {r->romS, mr, &e->S, r->vi, r->uj, us, w[0], w[1], w[2], w[3], 0, 0};
  if (r->romT && e->T.p) f[nf++] = {r->romT, mr, &e->T, r->vi, r->uj, us, w[0], w[1], w[2], w[3], 0, 0};
  size_t total = 0;
  for (int i = 0; i < nf; i++) { f[i].off = total; total += (f[i].count() + 3) & ~(size_t)3; }
  const size_t es = (r->dtype == LT_ROM_F32) ? 4 : 8;
  const size_t bytes = total * es;
  HIPCHK(e, hipSetDevice(e->device));
  // pinned double buffer and device staging sized for a whole-grid record of doubles
  const size_t cap = ((size_t)e->rho_nodes * (1 + 2 * (size_t)ws + 2 * (size_t)us) + ((size_t)e->u_nodes + e->v_nodes) * us + 32) * 8;
  if (e->pin_bytes < cap) {
    for (int i = 0; i < 2; i++) {
      if (e->pin_busy[i]) HIPCHK(e, hipEventSynchronize(e->pin_free[i]));
      e->pin_busy[i] = false;
      if (e->pin[i]) (void)hipHostFree(e->pin[i]);
      e->pin[i] = nullptr;
      if (hipHostMalloc(&e->pin[i], cap) != hipSuccess) return fail(e, LT_ERR_NOMEM, "pinned staging of %zu bytes", cap);
    }
    e->pin_bytes = cap;
    HIPCHK(e, hipStreamSynchronize(e->copy_stream));
    TRY(dev_alloc(e, e->stage_rom, cap));
  }
  const int b = e->pin_next;
  e->pin_next ^= 1;
  if (e->pin_busy[b]) HIPCHK(e, hipEventSynchronize(e->pin_free[b]));  // its last H2D copy
  // pack: a few host threads share the (field, level) planes
  {
    struct Job { int fi, k_lo, k_hi; };
    std::vector<Job> jobs;
    for (int i = 0; i < nf; i++) {
      const int step = std::max(1, f[i].L / 8);
      for (int k = 0; k < f[i].L; k += step) jobs.push_back({i, k, std::min(f[i].L, k + step)});
    }
    const int nt = (int)std::min<size_t>(8, std::max<size_t>(1, bytes >> 22));
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (size_t j; (j = next.fetch_add(1)) < jobs.size();)
        pack_field(f[jobs[j].fi], es, (char*)e->pin[b], jobs[j].k_lo, jobs[j].k_hi);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  // the spare slot may still be read (as the old back record) by steps enqueued before
  // the last commit
  if (e->mark_valid) HIPCHK(e, hipStreamWaitEvent(e->copy_stream, e->compute_mark, 0));
  HIPCHK(e, hipMemcpyAsync(e->stage_rom.p, e->pin[b], bytes, hipMemcpyHostToDevice, e->copy_stream));
  HIPCHK(e, hipEventRecord(e->pin_free[b], e->copy_stream));
  e->pin_busy[b] = true;
  const bool f32 = e->cfg.slab_storage == LT_SLAB_F32;
  for (int i = 0; i < nf; i++) {
    if (f32 && es == 4) launch_rom_scatter<float, float>(e, f[i], e->spare);
    else if (f32) launch_rom_scatter<float, double>(e, f[i], e->spare);
    else if (es == 4) launch_rom_scatter<double, float>(e, f[i], e->spare);
    else launch_rom_scatter<double, double>(e, f[i], e->spare);
  }
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipEventRecord(e->prefetch_done, e->copy_stream));
  e->prefetch_pending = true;
  return LT_OK;
}

int lt_slab_commit(lt_engine* e, int32_t t_b, int32_t t_c, int32_t t_f) {
  if (!e) return LT_ERR_ARG;
  if (e->pre_valid) return fail(e, LT_ERR_STATE, "not between lt_count_draws and its lt_step");
  if (t_b < 1 || t_b > 3 || t_c < 1 || t_c > 3 || t_f < 1 || t_f > 3 || t_b == t_c || t_c == t_f || t_b == t_f)
    return fail(e, LT_ERR_ARG, "time slots must be a permutation of 1..3");
  if (!e->prefetch_pending) return fail(e, LT_ERR_STATE, "lt_slab_commit without lt_slab_prefetch");
  HIPCHK(e, hipSetDevice(e->device));
  e->prefetch_pending = false;
  HIPCHK(e, hipEventSynchronize(e->prefetch_done));
  int flag = 0;
  HIPCHK(e, hipMemcpy(&flag, (int*)e->flags.p + 1, 4, hipMemcpyDeviceToHost));
  if (flag) {
    (void)hipMemset((int*)e->flags.p + 1, 0, 4);
    return fail(e, LT_ERR_SLAB_NOT_F32,
                "slab value is not exactly representable in the device storage type; "
                "use slab_storage = LT_SLAB_F64");
  }
  // later steps read the new record: ordered after the prefetch on the device as well
  HIPCHK(e, hipStreamWaitEvent(e->stream, e->prefetch_done, 0));
  const int old = e->phys[t_f];
  e->phys[t_f] = e->spare;
  e->spare = old;
  e->t_b = t_b; e->t_c = t_c; e->t_f = t_f;
  e->slab_set[t_f] = true;
  HIPCHK(e, hipEventRecord(e->compute_mark, e->stream));
  e->mark_valid = true;
  return LT_OK;
}

int lt_set_time_slots(lt_engine* e, int32_t t_b, int32_t t_c, int32_t t_f) {
  if (!e) return LT_ERR_ARG;
  if (e->pre_valid) return fail(e, LT_ERR_STATE, "not between lt_count_draws and its lt_step");
  if (t_b < 1 || t_b > 3 || t_c < 1 || t_c > 3 || t_f < 1 || t_f > 3)
    return fail(e, LT_ERR_ARG, "time slots must be 1..3");
  e->t_b = t_b; e->t_c = t_c; e->t_f = t_f;
  return LT_OK;
}

int lt_set_rng(lt_engine* e, const int32_t* mt624, int32_t mti) {
  if (!e || !mt624 || mti < 0 || mti > 624) return LT_ERR_ARG;
  for (int i = 0; i < 624; i++) e->mt[i] = (uint32_t)mt624[i];
  e->mti = mti;
  e->rng_ready = false;  // the generator states are re-positioned at the next use
  e->rng_cur = (uint64_t)mti;
  return LT_OK;
}

int lt_get_rng(lt_engine* e, int32_t* mt624, int32_t* mti) {
  if (!e || !mt624 || !mti) return LT_ERR_ARG;
  if (!e->rng_ready || e->rng_cur <= 624) {
    // nothing (or only words of the given array) consumed: the array is still the state
    for (int i = 0; i < 624; i++) mt624[i] = (int32_t)e->mt[i];
    *mti = (int32_t)(e->rng_ready ? e->rng_cur : (uint64_t)e->mti);
    return LT_OK;
  }
  // the 624 raw words before the cursor are the history window: as an mt array with
  // mti = 624 the reference's genrand regenerates exactly the words that follow
  HIPCHK(e, hipSetDevice(e->device));
  TRY(rng_ensure(e, e->rng_cur - 624, e->rng_cur - 1));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  const size_t ringw = (size_t)e->rng_R * RNG_S;
  for (int i = 0; i < 624; i++) {
    uint64_t k = e->rng_cur - 624 + i;
    uint32_t w;
    HIPCHK(e, hipMemcpy(&w, (uint32_t*)e->rng_ring.p + (k % ringw), 4, hipMemcpyDeviceToHost));
    mt624[i] = (int32_t)w;
  }
  *mti = 624;
  return LT_OK;
}

/* ---- test aids: only in the test build (-DLT_TEST_HOOKS -> libltrans_hip_test.so, tests/hip_test_lib.py); the
 * product library exports none of them ---- */
#ifdef LT_TEST_HOOKS
/* test aid (host only, no device): n tempered words of the stream that starts at
 * the given (mt, mti) state, beginning `offset` words after the next draw, computed
 * with the polynomial jump-ahead that positions the device generators. */
int lt_test_mt_stream(const int32_t* mt624, int32_t mti, uint64_t offset, int32_t n, uint32_t* out) {
  if (!mt624 || !out || mti < 0 || mti > 624 || n < 0) return LT_ERR_ARG;
  const ltmt::Poly& phi = mt_phi();
  if (phi.empty()) return LT_ERR_DEVICE;
  std::vector<uint32_t> h(624), w(624);
  for (int i = 0; i < 624; i++) h[i] = (uint32_t)mt624[i];
  uint64_t k = (uint64_t)mti + offset;  // virtual stream index of the first word wanted
  int i = 0;
  while (i < n && k < 624) out[i++] = ltmt::temper(h[k++]);
  if (i == n) return LT_OK;
  std::vector<uint32_t> g = ltmt::jump_polynomial(phi, k - 624);
  ltmt::host_jump(h.data(), g.data(), w.data());
  int p = 0;
  while (i < n) out[i++] = ltmt::temper(ltmt::host_next(w.data(), p));
  return LT_OK;
}

/* test aid (host only): 1 when x^a * x^b mod phi (poly_mulmod, how the engine forms the jump polynomials of a shard)
 * equals x^(a+b) mod phi formed from scratch, else 0. */
int lt_test_mt_mulmod(uint64_t a, uint64_t b) {
  const ltmt::Poly& phi = mt_phi();
  if (phi.empty()) return LT_ERR_DEVICE;
  const std::vector<uint32_t> ga = ltmt::jump_polynomial(phi, a), gb = ltmt::jump_polynomial(phi, b);
  return ltmt::poly_mulmod(phi, ga, gb) == ltmt::jump_polynomial(phi, a + b) ? 1 : 0;
}

/* test aid (device): q_fast[i] = div_prep(a[i], prep_div(b[i])), q_ref[i] = a[i] / b[i]
 * (host pointers), to check that the prepared-divisor division of lt_math.h is
 * bit-identical to the compiler's division on the operand ranges it is used for. */
__global__ void k_test_div(const double* a, const double* b, double* qf, double* qr, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ltd::Divisor d = ltd::prep_div(b[i]);
  qf[i] = ltd::div_prep(a[i], d);
  qr[i] = a[i] / b[i];
}
__global__ void k_test_div_by(const double* a, double b, double ch, double cl, double* qf, double* qr, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  qf[i] = ltd::div_by(a[i], b, ltd::DivConst{ch, cl}, true);
  qr[i] = a[i] / b;
}
/* test aid (device): div_by (lt_math.h) with the constants of lt_div_const against `/`, n numerators over one divisor;
 * LT_ERR_UNSUPPORTED when the divisor gets no constants */
int lt_test_div_by(const double* a, double b, double* q_fast, double* q_ref, int32_t n) {
  if (!a || !q_fast || !q_ref || n < 0) return LT_ERR_ARG;
  double hl[2];
  if (!lt_div_const(b, hl)) return LT_ERR_UNSUPPORTED;
  double *da = nullptr, *df = nullptr, *dr = nullptr;
  const size_t bytes = (size_t)n * 8;
  if (hipMalloc(&da, bytes) != hipSuccess || hipMalloc(&df, bytes) != hipSuccess || hipMalloc(&dr, bytes) != hipSuccess) return LT_ERR_NOMEM;
  (void)hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_test_div_by, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, b, hl[0], hl[1], df, dr, (int)n);
  const hipError_t rc = hipDeviceSynchronize();
  (void)hipMemcpy(q_fast, df, bytes, hipMemcpyDeviceToHost);
  (void)hipMemcpy(q_ref, dr, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(da); (void)hipFree(df); (void)hipFree(dr);
  return rc == hipSuccess ? LT_OK : LT_ERR_DEVICE;
}
int lt_test_div(const double* a, const double* b, double* q_fast, double* q_ref, int32_t n) {
  if (!a || !b || !q_fast || !q_ref || n < 0) return LT_ERR_ARG;
  double *da = nullptr, *db = nullptr, *df = nullptr, *dr = nullptr;
  const size_t bytes = (size_t)n * 8;
  if (hipMalloc(&da, bytes) != hipSuccess || hipMalloc(&db, bytes) != hipSuccess ||
      hipMalloc(&df, bytes) != hipSuccess || hipMalloc(&dr, bytes) != hipSuccess) return LT_ERR_NOMEM;
  (void)hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_test_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, df, dr, (int)n);
  const hipError_t rc = hipDeviceSynchronize();
  (void)hipMemcpy(q_fast, df, bytes, hipMemcpyDeviceToHost);
  (void)hipMemcpy(q_ref, dr, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(df); (void)hipFree(dr);
  return rc == hipSuccess ? LT_OK : LT_ERR_DEVICE;
}

/* test aid (device): the leaf numerics of lt_math.h on caller-supplied inputs, one item per
 * thread, so that tests can hold them against the vectors the REFERENCE's own modules produced
 * (tests/golden/leaf_golden.npz).  kind 1: polintd3, rows (xa1..3, ya1..3, x) -> y.
 * kind 2: gridcell_one, rows (ex1..4, ey1..4, x, y) -> 0/1.  kind 3: SNHCSH, x -> 3 values.
 * kind 4: a 4-knot spline, rows (x1..4, y1..4, t1..8) -> 37 values: the reference-shaped
 * YPC1 (4), SIGS tensions (3, zeros after a SigErr), SigErr count, HVAL and HPVAL at the 8
 * points (16); then what the fast path of k_rk computes for the same knots: YPC1 through the
 * prepared divisors and the branch-free sign-flip forms on the hardware min / max (4), tensions through sigs_classify +
 * sigs_newton / sigs_secant (3), SigErr flag, HVAL at the 8 points through hval_cubic_d (zero tension) or hval_interval_s (8;
 * the last 5 unused).  kind 5: x -> exp_core(x), lt_exp(x) (the host-libm-exact exp).  kind 6: x -> lt_log, lt_log10,
 * lt_pow_half, lt_cos, lt_sin, lt_sincos (sin, cos), lt_atan of lt_libm.h.  kind 7: the free-slip
 * substitution, rows (kind, water mask m, v-grid diagonal mask d, v1..v4) -> v1..v4.
 * kind 9: linint over n <= 124 points in the strided form of VTurb's SigErr fallback (linint_n) and, for
 * n = 4, the register form of k_rk_slow (linint4): rows (n, t, x[124], y[124]) -> value, slope, value4. */
__global__ void k_test_leaf(int kind, const double* in, double* out, int n) {
  using namespace ltd;
  libm_tables_load<LM_EXP | LM_LOG | LM_POW | LM_SINCOS>();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kind == 1) {
    const double* r = in + (size_t)i * 7;
    out[i] = polintd3(r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
  } else if (kind == 2) {
    const double* r = in + (size_t)i * 10;
    const double qx[4] = {r[0], r[1], r[2], r[3]}, qy[4] = {r[4], r[5], r[6], r[7]};
    out[i] = gridcell_one(qx, qy, r[8], r[9]) ? 1.0 : 0.0;
  } else if (kind == 3) {
    double a, b, c;
    snhcsh(in[i], a, b, c);
    out[3 * (size_t)i] = a; out[3 * (size_t)i + 1] = b; out[3 * (size_t)i + 2] = c;
  } else if (kind == 4) {
    const double* r = in + (size_t)i * 16;
    double* o = out + (size_t)i * 37;
    const double X[4] = {r[0], r[1], r[2], r[3]}, Y[4] = {r[4], r[5], r[6], r[7]};
    double YP[4], SG[3] = {0.0, 0.0, 0.0};
    int ier = 0, sigerr = 0;
    ypc1_4(X, Y, YP, ier);
    for (int iv = 0; iv < 3 && ier == 0; iv++) {
      const double sg = sigs_interval(X[iv + 1] - X[iv], Y[iv], Y[iv + 1], YP[iv], YP[iv + 1], sigerr);
      if (sigerr != 0) break;  // SIGS returns at the first failure
      SG[iv] = sg;
    }
    for (int k = 0; k < 4; k++) o[k] = YP[k];
    for (int k = 0; k < 3; k++) o[4 + k] = SG[k];
    o[7] = (double)sigerr;
    for (int k = 0; k < 8; k++) {
      const double T = r[8 + k];
      int I = (T < X[1]) ? 1 : (T < X[2]) ? 2 : 3;
      if (T > X[3]) I = 3;
      o[8 + k] = hval_interval(T, X[I - 1], X[I], Y[I - 1], Y[I], YP[I - 1], YP[I], SG[I - 1]);
      o[16 + k] = hpval_interval(T, X[I - 1], X[I], Y[I - 1], Y[I], YP[I - 1], YP[I], SG[I - 1]);
    }
    // the fast path's forms
    const double DX1 = X[1] - X[0], DX2 = X[2] - X[1], DX3 = X[3] - X[2];
    const Divisor d1 = prep_div(DX1), d2 = prep_div(DX2), d3 = prep_div(DX3);
    const Divisor d12 = prep_div(DX1 + DX2), d23 = prep_div(DX2 + DX3);
    double SL[3], YF[4], SF[3] = {0.0, 0.0, 0.0};
    SL[0] = div_prep(Y[1] - Y[0], d1); SL[1] = div_prep(Y[2] - Y[1], d2); SL[2] = div_prep(Y[3] - Y[2], d3);
    YF[0] = ypc1_end_b(SL[0], SL[0] + div_prep(DX1 * (SL[0] - SL[1]), d12));
    YF[1] = ypc1_mid_b(DX1, DX2, d12, SL[0], SL[1]);
    YF[2] = ypc1_mid_b(DX2, DX3, d23, SL[1], SL[2]);
    YF[3] = ypc1_end_b(SL[2], SL[2] + div_prep(DX3 * (SL[2] - SL[1]), d23));
    bool anyfail = false;
    for (int iv = 0; iv < 3; iv++) {
      double sg = 0.0, T = 0.0;
      bool failed = false;
      const int kind2 = sigs_classify(SL[iv], YF[iv], YF[iv + 1], sg, T);
      if (kind2 == SIGK_NEWTON) sg = sigs_newton(T, failed);
      else if (kind2 == SIGK_SECANT) sg = sigs_secant(SL[iv], YF[iv], YF[iv + 1]);
      SF[iv] = sg;
      anyfail = anyfail || failed;
    }
    for (int k = 0; k < 4; k++) o[24 + k] = YF[k];
    for (int k = 0; k < 3; k++) o[28 + k] = SF[k];
    o[31] = anyfail ? 1.0 : 0.0;
    for (int k = 0; k < 5; k++) {
      const double T = r[8 + k];
      int I = (T < X[1]) ? 1 : (T < X[2]) ? 2 : 3;
      if (T > X[3]) I = 3;
      const Divisor dI = (I == 1) ? d1 : (I == 2) ? d2 : d3;
      o[32 + k] = (SF[I - 1] == 0.0) ? hval_cubic_d(T, X[I - 1], dI, Y[I - 1], YF[I - 1], YF[I], SL[I - 1])
                                     : hval_interval_s(T, X[I - 1], X[I], Y[I - 1], YF[I - 1], YF[I], SL[I - 1], SF[I - 1]);
    }
  } else if (kind == 5) {
    out[2 * (size_t)i] = exp_core(in[i]);
    out[2 * (size_t)i + 1] = lt_exp(in[i]);
  } else if (kind == 6) {
    double* o = out + 8 * (size_t)i;
    const double x = in[i];
    o[0] = lt_log(x); o[1] = lt_log10(x); o[2] = lt_pow_half(x);
    o[3] = lt_cos(x); o[4] = lt_sin(x);
    lt_sincos(x, &o[5], &o[6]);
    o[7] = lt_atan_g(x);
  } else if (kind == 9) {
    const double* r = in + (size_t)i * 250;
    const int nn = (int)r[0];
    double y = 0.0, m = 0.0;
    linint_n(r + 2, r + 126, nn, 1, r[1], y, m);
    double* o = out + 3 * (size_t)i;
    o[0] = y; o[1] = m; o[2] = 0.0;
    if (nn == 4) {
      const double xa[4] = {r[2], r[3], r[4], r[5]}, ya[4] = {r[126], r[127], r[128], r[129]};
      o[2] = linint4(xa, ya, r[1]);
    }
  } else if (kind == 7) {  // free-slip substitution: rows (kind, m, d, v1..v4) -> v1..v4
    const double* r = in + (size_t)i * 7;
    double v1 = r[3], v2 = r[4], v3 = r[5], v4 = r[6];
    freeslip_sub((int)r[0], (unsigned)r[1], (unsigned)r[2], v1, v2, v3, v4);
    double* o = out + 4 * (size_t)i;
    o[0] = v1; o[1] = v2; o[2] = v3; o[3] = v4;
  }
}
int lt_test_leaf(int32_t kind, const double* in, int32_t n, int32_t in_per_item, double* out, int32_t out_per_item) {
  if (!in || !out || n < 0 || kind < 1 || kind > 9 || kind == 8) return LT_ERR_ARG;
  const int expect_in[10] = {0, 7, 10, 1, 16, 1, 1, 7, 0, 250}, expect_out[10] = {0, 1, 1, 3, 37, 2, 8, 4, 0, 3};
  if (in_per_item != expect_in[kind] || out_per_item != expect_out[kind]) return LT_ERR_ARG;
  double *di = nullptr, *dout = nullptr;
  const size_t bi = (size_t)n * in_per_item * 8, bo = (size_t)n * out_per_item * 8;
  if (hipMalloc(&di, bi + 8) != hipSuccess || hipMalloc(&dout, bo + 8) != hipSuccess) return LT_ERR_NOMEM;
  (void)hipMemcpy(di, in, bi, hipMemcpyHostToDevice);
  (void)hipMemset(dout, 0, bo);
  hipLaunchKernelGGL(k_test_leaf, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, (int)kind, di, dout, (int)n);
  const hipError_t rc = hipDeviceSynchronize();
  (void)hipMemcpy(out, dout, bo, hipMemcpyDeviceToHost);
  (void)hipFree(di); (void)hipFree(dout);
  return rc == hipSuccess ? LT_OK : LT_ERR_DEVICE;
}

/* test aid: the cursor of the engine's stream moved on by `nwords` as if that many words had been drawn (a run that
 * has been going for thousands of steps): the next draws are generated from there. */
int lt_test_rng_skip(lt_engine* e, uint64_t nwords) {
  if (!e) return LT_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->rng_ready) TRY(rng_init_device(e));
  e->rng_cur += nwords;
  return LT_OK;
}

/* test aid (device): n tempered words of the engine's stream starting `offset` words
 * after the cursor, read back from the materialised ring. */
int lt_test_mt_ring(lt_engine* e, uint64_t offset, int32_t n, uint32_t* out) {
  if (!e || !out || n < 0) return LT_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->rng_ready) TRY(rng_init_device(e));
  const uint64_t first = e->rng_cur + offset;
  TRY(rng_ensure(e, first, first + (uint64_t)n));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  const size_t ringw = (size_t)e->rng_R * RNG_S;
  std::vector<uint32_t> tmp(n);
  for (int i = 0; i < n;) {
    size_t pos = (size_t)((first + i) % ringw);
    size_t run = std::min<size_t>((size_t)(n - i), ringw - pos);
    HIPCHK(e, hipMemcpy(tmp.data() + i, (uint32_t*)e->rng_ring.p + pos, run * 4, hipMemcpyDeviceToHost));
    i += (int)run;
  }
  for (int i = 0; i < n; i++) out[i] = ltmt::temper(tmp[i]);
  return LT_OK;
}


/* test aid (device): inpoly (point_in_polygon_module.f90:25-167) for nq points against ONE closed polygon,
 * through the two forms the engine uses: the reference-shaped inpoly_full (onin = -1 absent / 1: on the
 * boundary counts as inside; 0: as outside) and the y-slab edge index of mbounds (lt_set_bounds builds it
 * with the same build_pip_index), which hands a point whose ray passes through a vertex to the full form.
 * out_full[q], out_index[q] (0/1; -1 where the indexed form does not apply: onin = 0), out_slow[q]. */
__global__ void k_test_inpoly(const StepParams P, const double* qx, const double* qy, const int* onin, int nq,
                              int* out_full, int* out_index, int* out_slow) {
  using namespace ltd;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  out_full[i] = inpoly_full(qx[i], qy[i], P.maxbound, P.bx, P.by, onin[i] == 0) ? 1 : 0;
  out_index[i] = -1; out_slow[i] = 0;
  if (onin[i] != 0) {
    bool slow = false;
    bool in = mbounds_indexed(P, qx[i], qy[i], slow);
    if (slow) in = inpoly_full(qx[i], qy[i], P.maxbound, P.bx, P.by, false);
    out_index[i] = in ? 1 : 0;
    out_slow[i] = slow ? 1 : 0;
  }
}
int lt_test_inpoly(const double* px, const double* py, int32_t npts, const double* qx, const double* qy,
                   const int32_t* onin, int32_t nq, int32_t* out_full, int32_t* out_index, int32_t* out_slow) {
  if (!px || !py || npts < 3 || !qx || !qy || !onin || nq < 0) return LT_ERR_ARG;
  PipIndex I; HostIndex H;
  if (!build_pip_index(px, py, nullptr, npts, [](int) { return true; }, I, H)) return LT_ERR_ARG;
  double *dpx, *dpy, *dqx, *dqy, *drec = nullptr, *dys = nullptr; int *dptr, *dids, *donin, *o1, *o2, *o3;
  if (hipMalloc(&drec, H.rec.size() * 8 + 32) != hipSuccess || hipMalloc(&dys, H.ys.size() * 8 + 32) != hipSuccess) return LT_ERR_NOMEM;
  (void)hipMemcpy(drec, H.rec.data(), H.rec.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dys, H.ys.data(), H.ys.size() * 8, hipMemcpyHostToDevice);
  const size_t nb = (size_t)npts * 8, qb = (size_t)nq * 8 + 8, ib = (size_t)nq * 4 + 4;
  if (hipMalloc(&dpx, nb) != hipSuccess || hipMalloc(&dpy, nb) != hipSuccess || hipMalloc(&dqx, qb) != hipSuccess ||
      hipMalloc(&dqy, qb) != hipSuccess || hipMalloc(&dptr, H.ptr.size() * 4 + 4) != hipSuccess ||
      hipMalloc(&dids, H.gidx.size() * 4 + 4) != hipSuccess || hipMalloc(&donin, ib) != hipSuccess ||
      hipMalloc(&o1, ib) != hipSuccess || hipMalloc(&o2, ib) != hipSuccess || hipMalloc(&o3, ib) != hipSuccess)
    return LT_ERR_NOMEM;
  (void)hipMemcpy(dpx, px, nb, hipMemcpyHostToDevice); (void)hipMemcpy(dpy, py, nb, hipMemcpyHostToDevice);
  (void)hipMemcpy(dqx, qx, (size_t)nq * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dqy, qy, (size_t)nq * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(donin, onin, (size_t)nq * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dptr, H.ptr.data(), H.ptr.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dids, H.gidx.data(), H.gidx.size() * 4, hipMemcpyHostToDevice);
  StepParams P;
  memset(&P, 0, sizeof(P));
  P.bx = dpx; P.by = dpy; P.maxbound = npts;
  P.pip_main = I; P.pip_main.ptr = dptr; P.pip_main.gidx = dids; P.pip_main.rec = (const double4*)drec; P.pip_main.pid = nullptr;
  P.pip_main.ys = dys;
  if (nq) hipLaunchKernelGGL(k_test_inpoly, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, 0, P, dqx, dqy, donin, (int)nq, o1, o2, o3);
  const hipError_t rc = hipDeviceSynchronize();
  (void)hipMemcpy(out_full, o1, (size_t)nq * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(out_index, o2, (size_t)nq * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(out_slow, o3, (size_t)nq * 4, hipMemcpyDeviceToHost);
  for (void* q : {(void*)dpx, (void*)dpy, (void*)dqx, (void*)dqy, (void*)dptr, (void*)dids, (void*)donin, (void*)o1, (void*)o2, (void*)o3, (void*)drec, (void*)dys}) (void)hipFree(q);
  return rc == hipSuccess ? LT_OK : LT_ERR_DEVICE;
}

/* test aid (HOST only, no device): the slab index of the point-in-polygon tests (lt_device.h: PipIndex) built as
 * lt_set_bounds builds it and queried by the very functions the kernels call (pip_slab, pip_edge: host-callable), for
 * one polygon (polyid = NULL: mbounds' form) or several (polyid(npts): ibounds' form, "inside any").  out_in[q] = 0/1
 * where the index decides, out_slow[q] = 1 where it hands over to the reference's full algorithm.  Returns the number
 * of list entries, or a negative status. */
int lt_test_pip_index_host(const double* px, const double* py, const int32_t* polyid, int32_t npts, const double* qx,
                           const double* qy, int32_t nq, int32_t* out_in, int32_t* out_slow) {
  if (!px || !py || npts < 2 || !qx || !qy || nq < 0 || !out_in || !out_slow) return LT_ERR_ARG;
  PipIndex I; HostIndex H;
  const bool ok = polyid ? build_pip_index(px, py, polyid, npts, [polyid](int i) { return polyid[i] == polyid[i + 1]; }, I, H)
                         : build_pip_index(px, py, nullptr, npts, [](int) { return true; }, I, H);
  if (!ok) return LT_ERR_UNSUPPORTED;
  std::vector<double4> rec4(H.ids.size() + 1);   // (double4's own alignment)
  for (size_t k = 0; k < H.ids.size(); k++) rec4[k] = make_double4(H.rec[4 * k], H.rec[4 * k + 1], H.rec[4 * k + 2], H.rec[4 * k + 3]);
  H.pid.resize(H.pid.size() + 1, 0);
  I.ys = H.ys.data(); I.gidx = H.gidx.data(); I.ptr = H.ptr.data(); I.rec = rec4.data(); I.pid = H.pid.data();
  for (int q = 0; q < nq; q++) {
    bool slow = false;
    const bool in = polyid ? ltd::pip_any_polygon(I, qx[q], qy[q], slow) : ltd::pip_one_polygon(I, qx[q], qy[q], slow);
    out_in[q] = in ? 1 : 0;
    out_slow[q] = slow ? 1 : 0;
  }
  return (int)H.ids.size();
}

/* test aid (device): N-knot tension splines (N = 4*ws in VTurb) through the forms the VTurb kernels use: the
 * streaming YPC1 of k_vt_table (ypc1_end / ypc1_mid in knot order), its screening of every interval
 * (sigs_eval_zero) with the task-list iteration of k_vt_sigs (sigs_classify + sigs_newton / sigs_secant), the
 * table layout [wave][knot][lane], and the walk's cached-interval evaluation (vt_seek + hval_prepared /
 * hpval_prepared) at 8 points per profile.  in: per profile (n, x[124], y[124], t[8]) = 257 doubles; out: per
 * profile yp[124], sigma[124], SigErr, hval[8], hpval[8] = 265 doubles. */
__global__ void k_test_spline_n(const double* in, double* out, ltd::VtRec* tab, double* sigp, int nprof, int maxn) {
  using namespace ltd;
  libm_tables_load<LM_EXP>();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nprof) return;
  const double* r = in + (size_t)i * 257;
  double* o = out + (size_t)i * 265;
  const int n = (int)r[0];
  const double *X = r + 1, *Y = r + 125, *T = r + 249;
  VtRec* t = tab + (size_t)(i >> 6) * (size_t)maxn * VT_TS + (size_t)(i & 63);
  double* sgp = sigp + (size_t)(i >> 6) * (size_t)maxn * VT_TS + (size_t)(i & 63);
  VtBits bits{0ull, 0ull, 0ull, 0ull};
  // YPC1 in the stream's order
  double Sp = 0.0, DXp = 0.0, YPp = 0.0;
  bool fail = false;
  for (int k = 1; k <= n; k++) { t[(size_t)(k - 1) * VT_TS] = VtRec{X[k - 1], Y[k - 1], 0.0}; sgp[(size_t)(k - 1) * VT_TS] = 0.0; }
  for (int k = 2; k <= n && !fail; k++) {
    const double DX = X[k - 1] - X[k - 2];
    if (DX <= 0.0) { fail = true; break; }
    const double S = div_prep(Y[k - 1] - Y[k - 2], prep_div(DX));
    const Divisor dsum = prep_div(DXp + DX);
    if (k == 3) { YPp = ypc1_end_b(Sp, Sp + div_prep(DXp * (Sp - S), dsum)); t[0].yp = YPp; }
    if (k >= 3) { const double YPc = ypc1_mid_b(DXp, DX, dsum, Sp, S); t[(size_t)(k - 2) * VT_TS].yp = YPc; YPp = YPc; }
    if (k == n) t[(size_t)(k - 1) * VT_TS].yp = ypc1_end_b(S, S + div_prep(DX * (S - Sp), dsum));
    Sp = S; DXp = DX;
  }
  int sigerr = 0;
  if (!fail) {
    for (int k = 0; k < n - 1 && !sigerr; k++) {   // SIGS returns at the first failure: later tensions stay 0
      const VtRec a = t[(size_t)k * VT_TS], b = t[(size_t)(k + 1) * VT_TS];
      const double S = (b.y - a.y) / (b.x - a.x);
      double sg = 0.0;
      if (!sigs_eval_zero(S, a.yp, b.yp)) {   // a listed interval: its bit, its factor in the plane
        double TT = 0.0;
        bool failed = false;
        const int kind = sigs_classify(S, a.yp, b.yp, sg, TT);
        if (kind == SIGK_NEWTON) sg = sigs_newton(TT, failed);
        else if (kind == SIGK_SECANT) sg = sigs_secant(S, a.yp, b.yp);
        if (failed) { sigerr = 1; sg = 0.0; }
        const unsigned long long bit = 1ull << (k & 63);
        if (k < 64) bits.w0 |= bit; else if (k < 128) bits.w1 |= bit; else if (k < 192) bits.w2 |= bit; else bits.w3 |= bit;
        sgp[(size_t)k * VT_TS] = sg;
      }
    }
  }
  for (int k = 0; k < n; k++) { o[k] = t[(size_t)k * VT_TS].yp; o[124 + k] = sgp[(size_t)k * VT_TS]; }
  o[248] = (double)sigerr;
  VtInterval ci;
  ci.I = 0; ci.X1 = ci.X2 = ci.Y1 = ci.Y2 = ci.P1 = ci.P2 = ci.SG = 0.0;
  ci.S = ci.D1 = ci.D2 = ci.SM = ci.CM = ci.CMM = 0.0; ci.dx.b = 1.0; ci.dx.y = 1.0;
  for (int j = 0; j < 8; j++) {
    vt_seek(ci, T[j], n, t, sgp, bits);
    o[249 + j] = hval_prepared(T[j], ci);
    o[257 + j] = hpval_prepared(T[j], ci);
  }
}
int lt_test_spline_n(const double* in, int32_t nprof, double* out) {
  if (!in || !out || nprof < 0) return LT_ERR_ARG;
  double *di, *dout; ltd::VtRec* tab;
  const size_t bi = (size_t)nprof * 257 * 8 + 8, bo = (size_t)nprof * 265 * 8 + 8;
  const size_t cells = (size_t)((nprof + 63) / 64) * 124 * 64;
  const size_t tb = cells * (sizeof(ltd::VtRec) + 8) + 32;
  if (hipMalloc(&di, bi) != hipSuccess || hipMalloc(&dout, bo) != hipSuccess || hipMalloc(&tab, tb) != hipSuccess) return LT_ERR_NOMEM;
  (void)hipMemcpy(di, in, (size_t)nprof * 257 * 8, hipMemcpyHostToDevice);
  (void)hipMemset(dout, 0, bo);
  if (nprof) hipLaunchKernelGGL(k_test_spline_n, dim3((unsigned)((nprof + 63) / 64)), dim3(64), 0, 0, di, dout, tab, (double*)(tab + cells), (int)nprof, 124);
  const hipError_t rc = hipDeviceSynchronize();
  (void)hipMemcpy(out, dout, (size_t)nprof * 265 * 8, hipMemcpyDeviceToHost);
  (void)hipFree(di); (void)hipFree(dout); (void)hipFree(tab);
  return rc == hipSuccess ? LT_OK : LT_ERR_DEVICE;
}
#endif  /* LT_TEST_HOOKS */

static int ready_to_step(lt_engine* e) {
  if (!e->have_grid || !e->have_particles) return fail(e, LT_ERR_STATE, "grid / particles missing");
  if (!e->slab_set[1] || !e->slab_set[2] || !e->slab_set[3]) return fail(e, LT_ERR_STATE, "three slabs must be loaded");
  if (!e->have_bounds) return fail(e, LT_ERR_STATE, "bounds missing");
  if (e->cfg.settlementon && !e->have_habitat) return fail(e, LT_ERR_STATE, "habitat missing");
  return LT_OK;
}

int lt_count_active(lt_engine* e, const double* ix3, int64_t* n_active) {
  if (!e || !ix3 || !n_active) return LT_ERR_ARG;
  if (!e->have_grid || !e->have_particles) return fail(e, LT_ERR_STATE, "grid / particles missing");
  HIPCHK(e, hipSetDevice(e->device));
  StepParams P;
  fill_params(e, P);
  P.ix0 = ix3[0]; P.ix1 = ix3[1]; P.ix2 = ix3[2];
  // 64 partial counters, one 64-byte line each
  if (e->count_slots.bytes < 64 * 64) TRY(dev_alloc(e, e->count_slots, 64 * 64));
  unsigned long long* slot = (unsigned long long*)e->count_slots.p;
  HIPCHK(e, hipMemsetAsync(slot, 0, 64 * 64, e->stream));
  unsigned blocks = (unsigned)((e->cfg.numpar + 255) / 256);
  if (blocks) hipLaunchKernelGGL(k_count_active, dim3(blocks), dim3(256), 0, e->stream, P, slot);
  HIPCHK(e, hipGetLastError());
  unsigned long long part[64 * 8];
  HIPCHK(e, hipMemcpyAsync(part, slot, 64 * 64, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  unsigned long long v = 0;
  for (int i = 0; i < 64; i++) v += part[i * 8];
  *n_active = (int64_t)v;
  return LT_OK;
}

static int end_static_window(lt_engine* e);
int lt_set_rng_window(lt_engine* e, int64_t active_before, int64_t active_total) {
  if (!e || active_before < 0 || active_total < active_before) return LT_ERR_ARG;
  if (e->cfg.Behavior >= 1 && e->cfg.Behavior <= 5)
    return fail(e, LT_ERR_ARG, "behaviours 1-5 draw a data-dependent number of deviates: use "
                               "lt_count_draws / lt_set_rng_window_draws");
  HIPCHK(e, hipSetDevice(e->device));
  const int64_t D = (e->cfg.HTurbOn ? 4 : 0) + (e->cfg.VTurbOn ? 2 * (e->cfg.idt / 2) : 0);
  e->rng_before = active_before * D;
  e->rng_total = active_total * D;
  return end_static_window(e);
}

// A static window ends (a window is set anew, or the per-step exchange takes over).  Whether it held is read from the
// device NOW (nobody may have asked yet), the word is cleared -- a window set anew: the report is spent -- and the
// switch is made; a window that did not hold is this call's error, once.
static int end_static_window(lt_engine* e) {
  unsigned long long missed = 0;
  const bool reported = e->miss_reported || (e->h_counters && e->h_counters[CNT_STATIC_MISS] != 0);  // (an earlier call has said so already)
  if (e->rng_static && !reported) {
    HIPCHK(e, hipMemcpyAsync(e->h_counters + CNT_STATIC_MISS, (unsigned long long*)e->counters.p + CNT_STATIC_MISS,
                             sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  e->rng_static = false;
  e->rng_static_own = -1;
  e->miss_inflight = false;
  e->miss_reported = false;
  if (e->h_counters && e->h_counters[CNT_STATIC_MISS]) {
    missed = e->h_counters[CNT_STATIC_MISS];
    e->h_counters[CNT_STATIC_MISS] = 0;
    HIPCHK(e, hipMemsetAsync((unsigned long long*)e->counters.p + CNT_STATIC_MISS, 0, sizeof(unsigned long long), e->stream));
  }
  if (missed && !reported && !e->quiet_window_switch)
    return fail(e, LT_ERR_STATE, "static random-stream window: the shard's draws changed in %llu step(s) before this call "
                                 "ended the window (the steps since then used wrong positions of the stream); the new window is set", missed);
  return LT_OK;
}

int lt_set_rng_window_static(lt_engine* e, int64_t active_before, int64_t active_total) {
  if (!e) return LT_ERR_ARG;
  e->quiet_window_switch = true;   // (setting a static window anew spends the report of the one before, as it always did)
  const int rc = lt_set_rng_window(e, active_before, active_total);
  e->quiet_window_switch = false;
  if (rc != LT_OK) return rc;
  e->rng_static = active_total > 0;
  e->rng_static_before = e->rng_before;
  e->rng_static_total = e->rng_total;
  e->rng_static_own = -1;
  HIPCHK(e, hipMemsetAsync((unsigned long long*)e->counters.p + CNT_STATIC_MISS, 0, sizeof(unsigned long long), e->stream));
  return LT_OK;
}

int lt_set_rng_window_draws(lt_engine* e, int64_t draws_before, int64_t draws_total) {
  if (!e || draws_before < 0 || draws_total < draws_before) return LT_ERR_ARG;
  e->rng_before = draws_before;
  e->rng_total = draws_total;
  return LT_OK;
}

// 1/b as the double-double div_by (lt_math.h) multiplies by: hl[0] + hl[1] = 1/b to 2^-105, both of b's sign.  Offered
// only for b = B * 2^e with an odd integer B < 2^30 and an exponent well inside the range -- the divisors of the path
// (4*ws, 6, differences of whole seconds) are such; anything else makes the kernels divide (fastdiv = 0).
int lt_div_const(double b, double* hl) {
  if (!hl) return 0;
  hl[0] = hl[1] = 0.0;
  if (!(std::fabs(b) > 0x1p-500 && std::fabs(b) < 0x1p500)) return 0;
  int ex;
  const double m = std::fabs(std::frexp(b, &ex));              // [0.5, 1)
  unsigned long long M = (unsigned long long)std::ldexp(m, 53);  // the 53-bit significand as an integer
  while (M && !(M & 1ull)) M >>= 1;
  if (M >= (1ull << 30)) return 0;
  double h = 1.0 / b;
  double l = std::fma(-b, h, 1.0) / b;                          // (1 - b h is exact: 1/b - h, divided)
  if (l != 0.0 && std::signbit(l) != std::signbit(h)) {        // h overshoots: one ulp towards zero, the rest is of h's sign
    h = std::nextafter(h, 0.0);
    l = std::fma(-b, h, 1.0) / b;
  }
  if (l == 0.0) l = std::copysign(0.0, h);
  hl[0] = h; hl[1] = l;
  return 1;
}

static int phase_flush(lt_engine* e) {
  if (!e->phase_used) return LT_OK;
  HIPCHK(e, hipEventSynchronize(e->phase_ev[e->phase_used - 1]));
  for (size_t s0 = 0; s0 + 5 <= e->phase_used; s0 += 5)
    for (int k = 0; k < 4; k++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e->phase_ev[s0 + k], e->phase_ev[s0 + k + 1]) == hipSuccess) e->phase_ms[k] += ms;
    }
  e->phase_used = 0;
  return LT_OK;
}
static int phase_mark(lt_engine* e, int k) {  // k = 0..4 within the current step
  if (!e->phase_on) return LT_OK;
  if (k == 0 && e->phase_used + 5 > e->phase_ev.size()) TRY(phase_flush(e));
  HIPCHK(e, hipEventRecord(e->phase_ev[e->phase_used + k], e->stream));
  if (k == 4) e->phase_used += 5;
  return LT_OK;
}

// The first part of a step: parameters, k_pre, the behaviour pre-pass and -- when anything
// draws -- the prefix sum that places every tracked particle in the serial random stream.
// Returns with e->stream idle and the shard's number of draws known.
static int step_prepare(lt_engine* e, const double* ex3, const double* ix3, int32_t p, int32_t it) {
  if (e->resort_every > 0 && ++e->steps_since_sort > e->resort_every) {
    TRY(resort(e));
    e->steps_since_sort = 0;
  }
  StepParams& P = e->pre_P;
  fill_params(e, P);
  P.ex0 = ex3[0]; P.ex1 = ex3[1]; P.ex2 = ex3[2];
  {
    const double dv[5] = {ex3[1] - ex3[2], ex3[0] - ex3[1], ex3[0] - ex3[2], (double)(4 * e->cfg.ws), 6.0};
    double* dst[5] = {P.c_d23, P.c_d12, P.c_d13, P.c_p2, P.c_6};
    int ok = 1;
    for (int i = 0; i < 5; i++) ok &= lt_div_const(dv[i], dst[i]) ? 1 : 0;
    P.fastdiv = (ok && !getenv("LT_NO_FASTDIV")) ? 1 : 0;
  }
  P.ix0 = ix3[0]; P.ix1 = ix3[1]; P.ix2 = ix3[2];
  P.p = p; P.it = it;
  {
    // polintd's choice of the closest node (interpolation_module.f90:85-92) depends on the
    // times only
    int ns[3];
    for (int i = 0; i < 3; i++) {
      const double x = ix3[i];
      ns[i] = 1;
      double dif = std::fabs(x - ex3[0]), dift = std::fabs(x - ex3[1]);
      if (dift < dif) { ns[i] = 2; dif = dift; }
      dift = std::fabs(x - ex3[2]);
      if (dift < dif) { ns[i] = 3; dif = dift; }
    }
    P.pns0 = ns[0]; P.pns1 = ns[1]; P.pns2 = ns[2];
  }
  // (the step's counters are cleared by k_pre, the first kernel of the step)
  const unsigned blocks = (unsigned)((e->cfg.numpar + 255) / 256);
  const bool beh_draws = e->cfg.Behavior >= 1 && e->cfg.Behavior <= 5;
  const bool beh_pre = beh_draws || e->cfg.Behavior == 7 || e->cfg.SaltTempOn;
  const bool draws = e->cfg.HTurbOn || e->cfg.VTurbOn || beh_draws;
  e->pre_own = 0;
  TRY(phase_mark(e, 0));
  if (blocks) {
    const bool f32 = e->cfg.slab_storage == LT_SLAB_F32;
    if (f32) hipLaunchKernelGGL((k_pre<float>), dim3(blocks), dim3(256), 0, e->stream, P);
    else hipLaunchKernelGGL((k_pre<double>), dim3(blocks), dim3(256), 0, e->stream, P);
    if (beh_pre) {
      if (f32) hipLaunchKernelGGL((k_beh_pre<float>), dim3(blocks), dim3(256), 0, e->stream, P);
      else hipLaunchKernelGGL((k_beh_pre<double>), dim3(blocks), dim3(256), 0, e->stream, P);
    }
    if (draws) {
      // offset of every tracked particle's first draw, in REFERENCE order: the reference's
      // generator is advanced particle by particle (LTRANS.f90:778,1087,1098,1110)
      const int n = e->cfg.numpar;
      hipStream_t ss = e->stream;
      hipLaunchKernelGGL(k_flag_scatter, dim3(blocks), dim3(256), 0, ss, (const int*)e->ws_i.p,
                         (const int*)e->orig.p, (int*)e->flag.p, n, P.rng_draws);
      size_t need = 0;
      HIPCHK(e, hipcub::DeviceScan::ExclusiveSum(nullptr, need, (int*)e->flag.p, (int*)e->rank.p, n + 1, ss));
      if (need > e->scan_tmp_bytes) {
        TRY(dev_alloc(e, e->scan_tmp, need));
        e->scan_tmp_bytes = e->scan_tmp.bytes;
      }
      HIPCHK(e, hipMemsetAsync((int*)e->flag.p + n, 0, 4, ss));
      size_t tb = e->scan_tmp_bytes;
      HIPCHK(e, hipcub::DeviceScan::ExclusiveSum(e->scan_tmp.p, tb, (int*)e->flag.p, (int*)e->rank.p, n + 1, ss));
      // rank[n] = draws of this shard.  Under a static window (lt_set_rng_window_static: no particle can stop
      // drawing) it is read back once; afterwards the host knows it, the step is enqueued without waiting for the
      // device, and k_post compares the scan's total with it (a mismatch is reported by lt_get_stats /
      // lt_get_particles: LT_ERR_STATE).
      if (e->rng_static && e->rng_static_own >= 0) {
        e->pre_own = e->rng_static_own;
        P.expect_own = (long long)e->rng_static_own;
        e->pre_synced = false;
      } else if (e->pre_defer_read) {
        // lt_exchange_rng_window: the total travels to the host together with the other shards' (one read)
        e->pre_own = -1;
        e->pre_synced = false;
      } else {
        // on its way to the host; whoever needs it first waits for it (resolve_own: lt_step after it has enqueued
        // k_rk, lt_count_draws at once)
        if (!e->h_own) {
          HIPCHK(e, hipHostMalloc((void**)&e->h_own, 64, hipHostMallocDefault));
          HIPCHK(e, hipEventCreateWithFlags(&e->own_ev, hipEventDisableTiming));
        }
        HIPCHK(e, hipMemcpyAsync(e->h_own, (int*)e->rank.p + n, 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipEventRecord(e->own_ev, e->stream));
        e->own_pending = true;
        e->pre_own = -1;
        e->pre_synced = false;
      }
    }
  }
  HIPCHK(e, hipGetLastError());
  TRY(phase_mark(e, 1));
  e->pre_valid = true;
  e->pre_sig[0] = ex3[0]; e->pre_sig[1] = ex3[1]; e->pre_sig[2] = ex3[2];
  e->pre_sig[3] = ix3[0]; e->pre_sig[4] = ix3[1]; e->pre_sig[5] = ix3[2];
  e->pre_sig[6] = (double)p; e->pre_sig[7] = (double)it;
  return LT_OK;
}

// The step's draw totals on the host: waits for whichever copy is on its way (the shard's own total, or the job's
// gathered counts of lt_exchange_rng_window).  Both events lie behind the prefix sum of THIS step on the engine's
// stream, so the step before has completed once they have (pre_synced).
static int resolve_own(lt_engine* e) {
  if (e->xchg_pending) {
    HIPCHK(e, hipEventSynchronize(e->cnt_arrived));
    e->xchg_pending = false;
    long long before = 0, total = 0;
    for (int r = 0; r < e->xchg_G; r++) {
      if (e->h_cnt[r] < 0) return fail(e, LT_ERR_STATE, "lt_exchange_rng_window: rank %d reports %lld draws", r, e->h_cnt[r]);
      if (r < e->xchg_rank) before += e->h_cnt[r];
      total += e->h_cnt[r];
    }
    e->pre_own = (int64_t)e->h_cnt[e->xchg_rank];
    e->rng_before = before;
    e->rng_total = total;
    e->pre_synced = true;
  } else if (e->own_pending) {
    HIPCHK(e, hipEventSynchronize(e->own_ev));
    e->own_pending = false;
    e->pre_own = (int64_t)*e->h_own;
    e->pre_synced = true;
    if (e->rng_static) e->rng_static_own = e->pre_own;
  }
  return LT_OK;
}

static bool pre_matches(const lt_engine* e, const double* ex3, const double* ix3, int32_t p, int32_t it) {
  return e->pre_valid && e->pre_sig[0] == ex3[0] && e->pre_sig[1] == ex3[1] && e->pre_sig[2] == ex3[2] &&
         e->pre_sig[3] == ix3[0] && e->pre_sig[4] == ix3[1] && e->pre_sig[5] == ix3[2] &&
         e->pre_sig[6] == (double)p && e->pre_sig[7] == (double)it;
}

int lt_count_draws(lt_engine* e, const double* ex3, const double* ix3, int32_t p, int32_t it, int64_t* n_draws) {
  if (!e || !ex3 || !ix3 || !n_draws) return LT_ERR_ARG;
  TRY(ready_to_step(e));
  HIPCHK(e, hipSetDevice(e->device));
  if (e->pre_valid) return fail(e, LT_ERR_STATE, "lt_count_draws: the step counted before has not been run");
  TRY(step_prepare(e, ex3, ix3, p, it));
  TRY(resolve_own(e));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  *n_draws = e->pre_own;
  return LT_OK;
}

int lt_step(lt_engine* e, const double* ex3, const double* ix3, int32_t p, int32_t it) {
  if (!e || !ex3 || !ix3) return LT_ERR_ARG;
  TRY(ready_to_step(e));
  HIPCHK(e, hipSetDevice(e->device));
  if (e->pre_valid && !pre_matches(e, ex3, ix3, p, it))
    return fail(e, LT_ERR_STATE, "lt_step: arguments differ from the preceding lt_count_draws");
  if (!e->pre_valid) TRY(step_prepare(e, ex3, ix3, p, it));
  e->pre_valid = false;
  StepParams& P = e->pre_P;
  const unsigned blocks = (unsigned)((e->cfg.numpar + 255) / 256);
  const bool draws = e->cfg.HTurbOn || e->cfg.VTurbOn || (e->cfg.Behavior >= 1 && e->cfg.Behavior <= 5);
  e->stats.rng_draws = 0;
  if (blocks) {
    const bool prof = e->prof_on && e->prof_used + 2 <= e->prof_ev.size();
    const bool f32 = e->cfg.slab_storage == LT_SLAB_F32;
    // the step's random words: where the shard's draws lie in the stream, and the generators of the blocks they touch
    // (on a stream of their own, beside k_rk)
    auto place_generators = [&]() -> int {
      const int64_t own = e->pre_own;
      if (own < 0) return fail(e, LT_ERR_STATE, "lt_step: the shard's draw total is not known");
      const int64_t before = (e->rng_total >= 0) ? e->rng_before : 0;
      const int64_t total = (e->rng_total >= 0) ? e->rng_total : own;
      if (before + own > total) return fail(e, LT_ERR_ARG, "lt_set_rng_window: window smaller than the shard's draws");
      if (!e->rng_ready) TRY(rng_init_device(e));
      const uint64_t base = e->rng_cur + (uint64_t)before;
      if (own > 0) {
        // the generators may overwrite old ring blocks while k_rk runs on the main stream: everything earlier on
        // e->stream has completed (somebody waited for this step's prefix sum), or -- static window, nobody waited --
        // the generator stream waits for the end of the step before
        if (!e->pre_synced && e->step_done_valid) HIPCHK(e, hipStreamWaitEvent(e->rng_stream, e->step_done, 0));
        TRY(rng_ensure(e, base, base + (uint64_t)own, e->rng_stream));
        HIPCHK(e, hipEventRecord(e->rng_done, e->rng_stream));
        e->rng_pending = true;
      }
      P.rng_ring = (const uint32_t*)e->rng_ring.p;
      P.rng_mask = (unsigned long long)e->rng_R * RNG_S - 1ull;
      P.rng_base = base;
      e->rng_cur += (uint64_t)total;
      e->rng_total = e->rng_static ? e->rng_static_total : -1;
      e->rng_before = e->rng_static ? e->rng_static_before : 0;
      e->stats.rng_draws = own;
      return LT_OK;
    };
    // The generators go first: their few large workgroups (four generator states each, a compute unit's worth of wave
    // slots) are placed while the device is still empty and then run beside k_rk.  Enqueued AFTER k_rk they wait until
    // its queue of single-wave workgroups has drained -- a slot k_rk frees is refilled by k_rk -- and the step's random
    // words arrive a kernel late (measured: behaviour 4 on 1M particles 1.45 -> 2.15 ms per step, configs[4]'s shape
    // 4.77 -> 4.85 ms).  So a step whose draw totals are still on their way to the host (no static window: the
    // shard's own total behind the prefix sum, or the job's counts behind lt_exchange_rng_window's all-gather) waits
    // for them here, before anything else of the step is enqueued: the one host wait of such a step.
    if (draws) {
      TRY(resolve_own(e));
      TRY(place_generators());
    }
    // lt_profile_* times the dominant kernel(s) of the configuration: the VTurb kernels when
    // vertical turbulence is on, k_rk otherwise
    const bool prof_rk = prof && !e->cfg.VTurbOn, prof_vt = prof && e->cfg.VTurbOn;
    if (prof_rk) HIPCHK(e, hipEventRecord(e->prof_ev[e->prof_used], e->stream));
    launch_rk(e, P, blocks);
    if (prof_rk) {
      HIPCHK(e, hipEventRecord(e->prof_ev[e->prof_used + 1], e->stream));
      e->prof_used += 2;
    }
    const unsigned slow_blocks = blocks < 256u ? blocks : 256u;
    if (f32) hipLaunchKernelGGL((k_rk_slow<float>), dim3(slow_blocks), dim3(256), 0, e->stream, P);
    else hipLaunchKernelGGL((k_rk_slow<double>), dim3(slow_blocks), dim3(256), 0, e->stream, P);
    TRY(phase_mark(e, 2));
    if (prof_vt) HIPCHK(e, hipEventRecord(e->prof_ev[e->prof_used], e->stream));
    if (e->cfg.VTurbOn) {
      TRY(launch_vturb_head(e, P, blocks));
      TRY(launch_vturb_tail(e, P, blocks));
    }
    if (prof_vt) {
      HIPCHK(e, hipEventRecord(e->prof_ev[e->prof_used + 1], e->stream));
      e->prof_used += 2;
    }
    TRY(phase_mark(e, 3));
    wait_rng(e);
    hipLaunchKernelGGL(k_post, dim3(blocks), dim3(256), 0, e->stream, P);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(e->step_done, e->stream));
    e->step_done_valid = true;
    TRY(phase_mark(e, 4));
  } else {
    // an empty shard draws nothing, but the job's stream moves on by the step's total: the shard hands back the same
    // generator state as the others (lt_get_rng)
    if (draws) TRY(resolve_own(e));
    if (draws && e->rng_total > 0) {
      if (!e->rng_ready) TRY(rng_init_device(e));
      e->rng_cur += (uint64_t)e->rng_total;
    }
    if (draws) {
      e->rng_total = e->rng_static ? e->rng_static_total : -1;
      e->rng_before = e->rng_static ? e->rng_static_before : 0;
    }
    if (e->phase_on)
      for (int k = 2; k <= 4; k++) TRY(phase_mark(e, k));
  }
  e->stats_valid = false;  // the step's counters stay on the device until somebody asks (fetch_counters)
  e->counters_fetched = false;
  return LT_OK;
}

static int fetch_counters(lt_engine* e) {
  if (!e->counters_fetched) {
    HIPCHK(e, hipMemcpyAsync(e->h_counters, e->counters.p, CNT_ALLOC * sizeof(unsigned long long),
                             hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->counters_fetched = true;
  }
  if (e->h_counters[CNT_STATIC_MISS]) e->miss_reported = true;
  if (e->h_counters[CNT_STATIC_MISS])   // (reported by every call until a window is set again)
    return fail(e, LT_ERR_STATE, "static random-stream window: the shard's draws changed (%llu step(s)); a particle "
                                 "stopped or started drawing -- set the window before each step instead",
                e->h_counters[CNT_STATIC_MISS]);
  return LT_OK;
}

int lt_sync(lt_engine* e) {
  if (!e) return LT_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return LT_OK;
}

int lt_get_stats(lt_engine* e, lt_step_stats* out) {
  if (!e || !out) return LT_ERR_ARG;
  TRY(lt_sync(e));
  TRY(fetch_counters(e));
  if (!e->stats_valid) {
    e->stats.active = (int64_t)e->h_counters[CNT_ACTIVE];
    e->stats.errors = (int64_t)e->h_counters[CNT_ERRORS];
    e->stats_valid = true;
  }
  *out = e->stats;
  return LT_OK;
}

/* diagnostics of the last step (the kernels' counters, lt_device.h: Counter): [0] tracked particles [1] particles
 * that raised an error [2] SigErr fallbacks of k_rk [3] brute-force boundary scans [4] full point-in-polygon
 * evaluations [5] VTurb spline intervals iterated [6] rounds of that iteration [7] particle-steps k_rk handed
 * to k_rk_slow.  (The re-sorts of the particle order are not per-step figures: lt_get_resort_stats.) */
#ifdef LT_RK_TIMING
int lt_diag(unsigned long long* out) {
  hipDeviceSynchronize();
  // (k_vt_table's sections are counted in this unit's copy of g_tim, k_rk's in lt_rk_kernel.hip's)
  hipMemcpyFromSymbol(out, HIP_SYMBOL(ltd::g_tim), 128);
  unsigned long long z[16] = {0}, r[16];
  hipMemcpyToSymbol(HIP_SYMBOL(ltd::g_tim), z, 128);
  ltd::rk_timing_fetch(r, true);
  if (!getenv("LT_DIAG_ENGINE_UNIT_ONLY"))   // (tools/dev/vt_walk_stats.py: slots 0..7 are k_vt_walk's there)
    for (int i = 0; i < 16; i++) out[i] += r[i];
  return 0;
}
#endif
int lt_get_counters(lt_engine* e, uint64_t* out8) {
  if (!e || !out8) return LT_ERR_ARG;
  TRY(lt_sync(e));
  TRY(fetch_counters(e));
  for (int i = 0; i < CNT_COUNT; i++) out8[i] = e->h_counters[i];
  return LT_OK;
}

int lt_get_resort_stats(lt_engine* e, int64_t* resorts, double* last_ms) {
  if (!e || !resorts || !last_ms) return LT_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  *resorts = (int64_t)e->resorts;
  *last_ms = -1.0;   // (no re-sort yet, or its events could not be read)
  if (e->resorts) {
    HIPCHK(e, hipStreamSynchronize(e->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->resort_ev[0], e->resort_ev[1]) == hipSuccess) *last_ms = (double)ms;
  }
  return LT_OK;
}

/* Per-phase GPU time of the steps (the counterpart of WriteModelTiming, LTRANS.f90:1389-1398, 1669-1689):
 * lt_phase_timing(e, 1) arms it (a few HIP events per step), lt_get_phase_times(e, ms4, reset) synchronises
 * and returns the accumulated milliseconds of: set-up (gates, setEle, setInterp, s-levels, draw offsets),
 * advection (RK4), vertical turbulence, update (HTurb, behaviour, reflection, boundaries, setEle, settlement,
 * commit). */
int lt_phase_timing(lt_engine* e, int32_t on) {
  if (!e) return LT_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (on && e->phase_ev.empty()) {
    e->phase_ev.resize(5 * 64);
    for (auto& ev : e->phase_ev) HIPCHK(e, hipEventCreate(&ev));
  }
  if (!on) TRY(phase_flush(e));
  e->phase_on = on != 0;
  return LT_OK;
}
int lt_get_phase_times(lt_engine* e, double* ms4, int32_t reset) {
  if (!e || !ms4) return LT_ERR_ARG;
  TRY(phase_flush(e));
  for (int k = 0; k < 4; k++) ms4[k] = e->phase_ms[k];
  if (reset) for (int k = 0; k < 4; k++) e->phase_ms[k] = 0.0;
  return LT_OK;
}

/* HIP-event timing of the step kernel on the engine's stream.
 * lt_profile_begin(e, max_steps) arms up to max_steps (start, stop) event pairs;
 * lt_profile_end(e, &total_ms, &launches) synchronises and sums them. */
int lt_profile_begin(lt_engine* e, int32_t max_steps) {
  if (!e || max_steps <= 0) return LT_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  while (e->prof_ev.size() < (size_t)max_steps * 2) {
    hipEvent_t ev;
    HIPCHK(e, hipEventCreate(&ev));
    e->prof_ev.push_back(ev);
  }
  e->prof_used = 0;
  e->prof_on = true;
  return LT_OK;
}

int lt_profile_end(lt_engine* e, double* total_ms, int32_t* launches) {
  if (!e || !total_ms || !launches) return LT_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  double sum = 0.0;
  for (size_t i = 0; i + 1 < e->prof_used; i += 2) {
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (int32_t)(e->prof_used / 2);
  e->prof_on = false;
  e->prof_used = 0;
  return LT_OK;
}


/* ---- multi-GPU: the end-of-step all-gather of the shards' (x,y,z), RCCL over xGMI, behind the ABI ---- */
namespace {
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // (the copy the process has already loaded, if any -- a host that also imports torch has torch's; the symbols
    // stay local to this handle)
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (api.h) break;
    }
    if (!api.h)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.h) break;
      }
    if (api.h) {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
      api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) api.h = nullptr;
    }
  }
  return api.h ? &api : nullptr;
}
// xyz_global(numpar_global,3) column-major <- the ranks' padded blocks [rank][3][nmax]
__global__ void k_unpad_positions(const double* __restrict__ recv, const long long* __restrict__ offs, int nranks,
                                  long long nmax, long long total, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int r = 0;
  while (r + 1 < nranks && offs[r + 1] <= i) r++;
  const long long j = i - offs[r];
  const double* b = recv + (size_t)r * 3 * (size_t)nmax;
  out[i] = b[j];
  out[(size_t)total + i] = b[(size_t)nmax + j];
  out[2 * (size_t)total + i] = b[2 * (size_t)nmax + j];
}
}  // namespace

int lt_comm_unique_id(void* id128) {
  if (!id128) return LT_ERR_ARG;
  RcclApi* R = rccl();
  if (!R) return LT_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return LT_ERR_DEVICE;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, 128);
  return LT_OK;
}

int lt_comm_init(lt_engine* e, const void* id128, int32_t nranks, int32_t rank) {
  if (!e || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return LT_ERR_ARG;
  if (e->comm) return fail(e, LT_ERR_STATE, "lt_comm_init: a communicator exists (lt_comm_destroy first)");
  RcclApi* R = rccl();
  if (!R) return fail(e, LT_ERR_UNSUPPORTED, "librccl.so could not be opened");
  HIPCHK(e, hipSetDevice(e->device));
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  const ncclResult_t rc = R->CommInitRank(&e->comm, nranks, id, rank);
  if (rc != ncclSuccess) {
    e->comm = nullptr;
    return fail(e, LT_ERR_DEVICE, "ncclCommInitRank: %s", R->GetErrorString ? R->GetErrorString(rc) : "error");
  }
  e->comm_nranks = nranks;
  e->comm_rank = rank;
  return LT_OK;
}

// The id file is good for ONE communicator: rank 0 removes whatever lies at `path` before it writes, and removes
// its own file once ncclCommInitRank has returned (every rank has joined by then, so every rank has read it);
// the other ranks ignore a file last written more than timeout_s (+ 2 s) before THIS CALL -- what an earlier or a
// crashed attempt left behind, also one of this very process -- and keep waiting for rank 0's.  Ranks are
// therefore expected to make the call within timeout_s of each other.  A fixed path shared by successive runs is
// still not safe (a rank that polls before rank 0 has removed the file of a run that crashed seconds ago reads a
// dead id and waits in ncclCommInitRank): use a path that is unique per run, as INTEGRATION.md says.
int lt_comm_init_file(lt_engine* e, const char* path, int32_t nranks, int32_t rank, int32_t timeout_s) {
  if (!e || !path || nranks < 1 || rank < 0 || rank >= nranks) return LT_ERR_ARG;
  unsigned char id[128];
  if (rank == 0) {
    TRY(lt_comm_unique_id(id));
    (void)unlink(path);
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, 128, f) != 128) { if (f) fclose(f); return fail(e, LT_ERR_ARG, "lt_comm_init_file: cannot write %s", tmp.c_str()); }
    fclose(f);
    if (rename(tmp.c_str(), path) != 0) return fail(e, LT_ERR_ARG, "lt_comm_init_file: cannot rename to %s", path);
  } else {
    bool ok = false;
    const time_t oldest = time(nullptr) - (time_t)(timeout_s > 0 ? timeout_s : 0) - 2;
    for (int waited_ms = 0; waited_ms <= timeout_s * 1000 && !ok; waited_ms += 50) {
      struct stat sb;
      if (stat(path, &sb) == 0 && sb.st_size == 128 && sb.st_mtime >= oldest) {
        FILE* f = fopen(path, "rb");
        if (f) { ok = fread(id, 1, 128, f) == 128; fclose(f); }
      }
      if (!ok) usleep(50 * 1000);
    }
    if (!ok) return fail(e, LT_ERR_STATE, "lt_comm_init_file: %s did not appear within %d s (a file left by an earlier run is ignored)", path, (int)timeout_s);
  }
  const int rc = lt_comm_init(e, id, nranks, rank);
  if (rank == 0) (void)unlink(path);
  return rc;
}

int lt_comm_destroy(lt_engine* e) {
  if (!e) return LT_ERR_ARG;
  if (e->comm) {
    RcclApi* R = rccl();
    (void)hipSetDevice(e->device);
    if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (R) (void)R->CommDestroy(e->comm);
    e->comm = nullptr;
  }
  return LT_OK;
}

namespace {
// The all-gather in three parts, so that the test build can stand in for the collective on one GPU:
//   ag_pack     sizes, buffers, this shard packed as [3][nmax] into the send buffer (engine's stream), and the
//               hand-over to the communication stream
//   (collective: ncclAllGather of the padded shards into [rank][3][nmax] on the communication stream)
//   ag_finish   the padded blocks assembled as xyz(sum(counts),3) in global particle order, the copy to the host
// The collective runs on its OWN stream, ordered by events: it waits for the pack; the engine's stream does not
// wait for it, so the next lt_step runs beside the transfer, and only the next pack (which overwrites the send
// buffer) waits for the previous collective.
struct AgPlan { int G = 0; long long nmax = 0, total = 0; double* out = nullptr; };
// every collective of the engine's communicator runs on ONE stream of its own, ordered against the engine's by events
int ensure_comm_stream(lt_engine* e) {
  if (!e->comm_stream) {
    HIPCHK(e, hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
    HIPCHK(e, hipEventCreateWithFlags(&e->comm_packed, hipEventDisableTiming));
    HIPCHK(e, hipEventCreateWithFlags(&e->comm_done, hipEventDisableTiming));
  }
  return LT_OK;
}
int ag_pack(lt_engine* e, int G, int rank, const int32_t* counts, void* xyz_device, AgPlan& A) {
  if (counts[rank] != e->cfg.numpar) return fail(e, LT_ERR_ARG, "lt_allgather_positions: counts(rank+1) /= numpar of this engine");
  A.G = G;
  std::vector<long long> offs(G + 1, 0);
  for (int r = 0; r < G; r++) {
    if (counts[r] < 0) return LT_ERR_ARG;
    A.nmax = std::max<long long>(A.nmax, counts[r]);
    offs[r + 1] = offs[r] + counts[r];
  }
  A.total = offs[G];
  if (A.total == 0) return LT_OK;
  TRY(ensure_comm_stream(e));
  // the buffers of the collective before this one are free once it has finished (also before they are re-allocated)
  if (e->comm_busy) HIPCHK(e, hipStreamWaitEvent(e->stream, e->comm_done, 0));
  const size_t blk = (size_t)3 * (size_t)A.nmax * 8;
  if (e->comm_send.bytes < blk || e->comm_recv.bytes < blk * G || e->comm_counts.bytes < (size_t)(G + 1) * 8 ||
      (!xyz_device && e->comm_out.bytes < (size_t)3 * A.total * 8)) {
    if (e->comm_busy) HIPCHK(e, hipEventSynchronize(e->comm_done));
    if (e->comm_send.bytes < blk) TRY(dev_alloc(e, e->comm_send, blk));
    if (e->comm_recv.bytes < blk * G) TRY(dev_alloc(e, e->comm_recv, blk * G));
    if (e->comm_counts.bytes < (size_t)(G + 1) * 8) TRY(dev_alloc(e, e->comm_counts, (size_t)(G + 1) * 8));
    if (!xyz_device && e->comm_out.bytes < (size_t)3 * A.total * 8) TRY(dev_alloc(e, e->comm_out, (size_t)3 * A.total * 8));
  }
  A.out = xyz_device ? (double*)xyz_device : (double*)e->comm_out.p;
  HIPCHK(e, hipMemcpyAsync(e->comm_counts.p, offs.data(), (size_t)(G + 1) * 8, hipMemcpyHostToDevice, e->stream));
  const int n = e->cfg.numpar;
  if (n) {
    // this shard as [3][nmax]: xyz(numpar,3) column-major with the columns nmax apart
    const unsigned blocks = (unsigned)(((size_t)n + 255) / 256);
    hipLaunchKernelGGL(k_pack_positions, dim3(blocks), dim3(256), 0, e->stream, (const double*)e->par.p,
                       (const int*)e->orig.p, (double*)e->comm_send.p, n, (size_t)A.nmax);
    HIPCHK(e, hipGetLastError());
  }
  HIPCHK(e, hipEventRecord(e->comm_packed, e->stream));
  HIPCHK(e, hipStreamWaitEvent(e->comm_stream, e->comm_packed, 0));
  return LT_OK;
}
int ag_finish(lt_engine* e, const AgPlan& A, double* xyz_host) {
  hipLaunchKernelGGL(k_unpad_positions, dim3((unsigned)((A.total + 255) / 256)), dim3(256), 0, e->comm_stream,
                     (const double*)e->comm_recv.p, (const long long*)e->comm_counts.p, A.G, A.nmax, A.total, A.out);
  HIPCHK(e, hipGetLastError());
  if (xyz_host) HIPCHK(e, hipMemcpyAsync(xyz_host, A.out, (size_t)3 * A.total * 8, hipMemcpyDeviceToHost, e->comm_stream));
  HIPCHK(e, hipEventRecord(e->comm_done, e->comm_stream));
  e->comm_busy = true;
  if (xyz_host) HIPCHK(e, hipStreamSynchronize(e->comm_stream));
  return LT_OK;
}
}  // namespace

int lt_allgather_positions(lt_engine* e, const int32_t* counts, double* xyz_host, void* xyz_device) {
  if (!e || !counts || (!xyz_host && !xyz_device)) return LT_ERR_ARG;
  if (!e->comm) return fail(e, LT_ERR_STATE, "lt_allgather_positions before lt_comm_init");
  if (!e->have_particles) return fail(e, LT_ERR_STATE, "no particles uploaded");
  RcclApi* R = rccl();
  if (!R) return LT_ERR_UNSUPPORTED;
  HIPCHK(e, hipSetDevice(e->device));
  TRY(check_static_miss_async(e));
  AgPlan A;
  TRY(ag_pack(e, e->comm_nranks, e->comm_rank, counts, xyz_device, A));
  if (A.total == 0) return LT_OK;
  const ncclResult_t rc = R->AllGather(e->comm_send.p, e->comm_recv.p, (size_t)3 * (size_t)A.nmax, ncclDouble, e->comm, e->comm_stream);
  if (rc != ncclSuccess) return fail(e, LT_ERR_DEVICE, "ncclAllGather: %s", R->GetErrorString ? R->GetErrorString(rc) : "error");
  return ag_finish(e, A, xyz_host);
}

int lt_allgather_wait(lt_engine* e) {
  if (!e) return LT_ERR_ARG;
  if (e->comm_busy) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipEventSynchronize(e->comm_done));
  }
  return LT_OK;
}

/* ---- the per-step exchange of the shards' draw counts (SURVEY 8e's second collective) -------------------------
 * LTRANS draws from ONE serial stream in particle order (random_module.f90:166-197) and a particle that is not
 * tracked in a step draws nothing (LTRANS.f90:790-816), so a shard's place in the step's part of the stream is the
 * number of draws of the lower-ranked shards -- which changes whenever a particle anywhere settles, dies, leaves
 * or is released.  exchange_window: the first part of the coming step on the device (the gates, the element
 * search, the behaviour pre-pass, the prefix sum of draws: what lt_count_draws runs), the shard's total as one
 * int64 into the send word ON THE DEVICE, ncclAllGather of one int64 per rank on the communicator's stream behind
 * an event, and ONE read of the G counts by the host -- which needs them to plan the generators of the step
 * (rng_ensure) and which a non-static window costs a single engine as well (step_prepare's read of its own total:
 * folded into this one).  `injected` (test build): what the collective would deliver for the other ranks. */
namespace {
__global__ void k_own_draws_i64(const int* __restrict__ scan_total, long long* __restrict__ cnt) { cnt[0] = (long long)scan_total[0]; }

int exchange_window(lt_engine* e, const double* ex3, const double* ix3, int32_t p, int32_t it, int G, int rank,
                    bool collective, const int64_t* injected) {
  TRY(ready_to_step(e));
  HIPCHK(e, hipSetDevice(e->device));
  if (e->pre_valid && !pre_matches(e, ex3, ix3, p, it))
    return fail(e, LT_ERR_STATE, "lt_exchange_rng_window: arguments differ from the preceding lt_count_draws");
  if (e->rng_static) TRY(end_static_window(e));   // (the exchange takes over from a static window; a miss is reported here, once)
  const bool draws = e->cfg.HTurbOn || e->cfg.VTurbOn || (e->cfg.Behavior >= 1 && e->cfg.Behavior <= 5);
  if (!e->pre_valid) {
    e->pre_defer_read = draws && e->cfg.numpar > 0;
    const int rc = step_prepare(e, ex3, ix3, p, it);
    e->pre_defer_read = false;
    if (rc) return rc;
  }
  if (!draws) return LT_OK;                       // (nothing draws: there is no stream to place the shard in)
  TRY(ensure_comm_stream(e));
  if (!e->cnt_ready) HIPCHK(e, hipEventCreateWithFlags(&e->cnt_ready, hipEventDisableTiming));
  if (e->cnt_dev.bytes < (size_t)(G + 1) * 8) {
    if (e->comm_busy) HIPCHK(e, hipEventSynchronize(e->comm_done));
    TRY(dev_alloc(e, e->cnt_dev, (size_t)(G + 1) * 8));
  }
  if (e->h_cnt_cap < G) {
    if (e->h_cnt) (void)hipHostFree(e->h_cnt);
    e->h_cnt = nullptr; e->h_cnt_cap = 0;
    HIPCHK(e, hipHostMalloc((void**)&e->h_cnt, (size_t)G * 8, hipHostMallocDefault));
    e->h_cnt_cap = G;
  }
  long long* cnt = (long long*)e->cnt_dev.p;
  const int n = e->cfg.numpar;
  if (n > 0) hipLaunchKernelGGL(k_own_draws_i64, dim3(1), dim3(1), 0, e->stream, (const int*)e->rank.p + n, cnt);
  else HIPCHK(e, hipMemsetAsync(cnt, 0, 8, e->stream));          // (an empty shard draws nothing)
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipEventRecord(e->cnt_ready, e->stream));
  HIPCHK(e, hipStreamWaitEvent(e->comm_stream, e->cnt_ready, 0));
  if (collective) {
    RcclApi* R = rccl();
    if (!R) return fail(e, LT_ERR_UNSUPPORTED, "librccl.so could not be opened");
    const ncclResult_t rc = R->AllGather(cnt, cnt + 1, 1, ncclInt64, e->comm, e->comm_stream);
    if (rc != ncclSuccess) return fail(e, LT_ERR_DEVICE, "ncclAllGather (draw counts): %s", R->GetErrorString ? R->GetErrorString(rc) : "error");
  } else {
    // one rank, or the test build standing in for the collective: the other ranks' counts as given (through the pinned
    // words the result comes back to), this rank's own from the device
    if (injected && G > 1) {
      for (int r = 0; r < G; r++) e->h_cnt[r] = (long long)injected[r];
      HIPCHK(e, hipMemcpyAsync(cnt + 1, e->h_cnt, (size_t)G * 8, hipMemcpyHostToDevice, e->comm_stream));
    }
    HIPCHK(e, hipMemcpyAsync(cnt + 1 + rank, cnt, 8, hipMemcpyDeviceToDevice, e->comm_stream));
  }
  // The counts travel to the host behind the collective; nobody waits here: the lt_step that follows waits for them
  // (resolve_own) before it places the generators of the step's random words -- one host wait for the exchange and the
  // shard's own total together.
  if (!e->cnt_arrived) HIPCHK(e, hipEventCreateWithFlags(&e->cnt_arrived, hipEventDisableTiming));
  HIPCHK(e, hipMemcpyAsync(e->h_cnt, cnt + 1, (size_t)G * 8, hipMemcpyDeviceToHost, e->comm_stream));
  HIPCHK(e, hipEventRecord(e->cnt_arrived, e->comm_stream));
  e->xchg_pending = true;
  e->own_pending = false;     // (the shard's total comes with the others')
  e->xchg_G = G;
  e->xchg_rank = rank;
  e->pre_own = -1;
  return LT_OK;
}
}  // namespace

int lt_exchange_rng_window(lt_engine* e, const double* ex3, const double* ix3, int32_t p, int32_t it) {
  if (!e || !ex3 || !ix3) return LT_ERR_ARG;
  if (!e->comm) return exchange_window(e, ex3, ix3, p, it, 1, 0, false, nullptr);   // one engine: its own count
  return exchange_window(e, ex3, ix3, p, it, e->comm_nranks, e->comm_rank, true, nullptr);
}

#ifdef LT_TEST_HOOKS
/* Test aid: lt_exchange_rng_window as rank `rank` of `nranks` on ONE GPU -- counts(nranks) is what the collective
 * would deliver (the entry of this rank is ignored: its count comes from the device, as in the real call). */
int lt_test_exchange_rng_window(lt_engine* e, const double* ex3, const double* ix3, int32_t p, int32_t it, int32_t nranks,
                                int32_t rank, const int64_t* counts) {
  if (!e || !ex3 || !ix3 || !counts || nranks < 1 || rank < 0 || rank >= nranks) return LT_ERR_ARG;
  return exchange_window(e, ex3, ix3, p, it, nranks, rank, false, counts);
}
#endif

#ifdef LT_TEST_HOOKS
/* Test aid (libltrans_hip_test.so only): the N-rank form of lt_allgather_positions on ONE GPU.  This engine is
 * rank `rank` of `nranks`; the collective is replaced by what it would deliver: the receive buffer
 * [rank][3][nmax] holds this engine's own padded block (copied from its send buffer, as ncclAllGather does) and,
 * for every other rank r, counts[r] rows of others(sum of the other counts, 3) -- column-major, the other ranks'
 * particles in rank order -- padded to nmax; the padding is filled with NaN so that a row taken from it shows. */
int lt_test_allgather_layout(lt_engine* e, int32_t nranks, int32_t rank, const int32_t* counts, const double* others,
                             double* xyz_host) {
  if (!e || !counts || !xyz_host || nranks < 1 || rank < 0 || rank >= nranks) return LT_ERR_ARG;
  if (!e->have_particles) return fail(e, LT_ERR_STATE, "no particles uploaded");
  HIPCHK(e, hipSetDevice(e->device));
  AgPlan A;
  TRY(ag_pack(e, nranks, rank, counts, nullptr, A));
  if (A.total == 0) return LT_OK;
  long long n_others = 0;
  for (int r = 0; r < nranks; r++) if (r != rank) n_others += counts[r];
  std::vector<double> recv((size_t)nranks * 3 * (size_t)A.nmax, std::nan(""));
  long long o = 0;
  for (int r = 0; r < nranks; r++) {
    if (r == rank) continue;
    for (int c = 0; c < 3; c++)
      for (long long j = 0; j < counts[r]; j++)
        recv[((size_t)r * 3 + c) * (size_t)A.nmax + j] = others[(size_t)c * n_others + o + j];
    o += counts[r];
  }
  HIPCHK(e, hipMemcpyAsync(e->comm_recv.p, recv.data(), recv.size() * 8, hipMemcpyHostToDevice, e->comm_stream));
  HIPCHK(e, hipMemcpyAsync((double*)e->comm_recv.p + (size_t)rank * 3 * (size_t)A.nmax, e->comm_send.p,
                           (size_t)3 * (size_t)A.nmax * 8, hipMemcpyDeviceToDevice, e->comm_stream));
  const int rc = ag_finish(e, A, xyz_host);
  HIPCHK(e, hipStreamSynchronize(e->comm_stream));   // (recv is a local)
  return rc;
}
#endif


/* createBounds (boundary_module.f90:82-1215) on the host side of the library: lt_bounds_host.h does the walk,
 * this is the re-formatting of :1030-1204 into the arrays lt_set_bounds takes.  No device is touched. */
int lt_create_bounds(int32_t vi, int32_t uj, const int32_t* mask_rho, const double* x_u, const double* y_u,
                     const double* x_v, const double* y_v, int32_t cap, int32_t* nbounds_o, int32_t* maxbound_o,
                     int32_t* maxisland_o, int32_t* numislands_o, double* bx, double* by, double* hx, double* hy,
                     int32_t* hid, double* bnd_x, double* bnd_y, int32_t* land) {
  if (vi < 5 || uj < 5 || !mask_rho || !x_u || !y_u || !x_v || !y_v || !nbounds_o || !maxbound_o || !maxisland_o ||
      !numislands_o || !bx || !by || !hx || !hy || !hid || !bnd_x || !bnd_y || !land)
    return LT_ERR_ARG;
  return ltb::create_bounds_arrays(vi, uj, mask_rho, x_u, y_u, x_v, y_v, cap, nbounds_o, maxbound_o, maxisland_o, numislands_o,
                                  bx, by, hx, hy, hid, bnd_x, bnd_y, land) == 0 ? LT_OK : LT_ERR_ARG;
}

int lt_pack_positions(lt_engine* e, void* dst_device) {
  if (!e || !dst_device) return LT_ERR_ARG;
  if (!e->have_particles) return fail(e, LT_ERR_STATE, "no particles uploaded");
  HIPCHK(e, hipSetDevice(e->device));
  TRY(check_static_miss_async(e));
  unsigned blocks = (unsigned)(((size_t)e->cfg.numpar + 255) / 256);
  if (blocks)
    hipLaunchKernelGGL(k_pack_positions, dim3(blocks), dim3(256), 0, e->stream,
                       (const double*)e->par.p, (const int*)e->orig.p, (double*)dst_device,
                       e->cfg.numpar, (size_t)e->cfg.numpar);
  HIPCHK(e, hipGetLastError());
  return LT_OK;
}

void* lt_stream(lt_engine* e) { return e ? (void*)e->stream : nullptr; }

}  // extern "C"
