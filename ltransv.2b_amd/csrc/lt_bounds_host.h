// lt_bounds_host.h -- createBounds (boundary_module.f90:82-1215) on the host side of the library.
//
// The land / water boundary of the rho mask as closed polylines through the u and v points between rho
// points of different mask: the main polygon, clockwise round the water, then every island, in the ORDER
// the reference's walk produces them -- the arrays bx/by, hx/hy/hid, bnd_x/bnd_y, land are hot-path inputs
// (intersect_reflect's "first segment wins" tie-break, mbounds / ibounds) and their order is the walk's.
// It is a serial walk, O(boundary points); it stays on the host.
//
// Shape: the reference spells every element form out in SELECT CASE blocks, twice (getNext for the probes
// that resolve "cross" elements, and the walk itself with its calls to add).  Here the forms are a table
// (exit side per form), the four directions one rotation-symmetric rule for the strip of elements along the
// rim of the grid, and the probe and the walk share the interior rule.  What is kept literally, because
// results depend on it: the probe turns the corner of the rim strip one element earlier than the walk does
// (getNext :1300-1345 tests i == 2 / j == 2 / j == uj-2 / i == vi-2 where the walk :655-703 tests 1 / 1 /
// uj-1 / vi-1), crosses on the rim take forms 17 / 18, unresolved crosses default to 16 / 19, and the
// search orders for the start elements.
#pragma once
#include <cstdint>
#include <vector>

namespace ltb {

enum Side { BOTTOM = 0, LEFT = 1, RIGHT = 2, TOP = 3, NONE = -1 };
// direction codes of the reference (numeric keypad): 2 down (i-1), 4 left (j-1), 6 right (j+1), 8 up (i+1)
static const int DIR_OF_SIDE[4] = {2, 4, 6, 8};

// exit side of the clockwise-round-water walk per element form (1 = node (i,j+1) is land, 2 = (i,j), 4 = (i+1,j+1),
// 8 = (i+1,j)); 6 and 9 are the crosses, resolved into 16..19
static const signed char EXIT_WATER[16] = {NONE, BOTTOM, LEFT, LEFT, RIGHT, BOTTOM, NONE, LEFT,
                                           TOP, NONE, TOP, TOP, RIGHT, BOTTOM, RIGHT, NONE};
// ... and of the walk round an island (land on the right hand instead of water)
static const signed char EXIT_ISLAND[16] = {NONE, RIGHT, BOTTOM, RIGHT, TOP, TOP, NONE, TOP,
                                            LEFT, NONE, BOTTOM, RIGHT, LEFT, LEFT, BOTTOM, NONE};

struct Point { bool onU; int i, j, poly; };

class Bounds {
 public:
  Bounds(int vi, int uj, const int32_t* mask_rho) : vi_(vi), uj_(uj), m_((size_t)vi * uj), form_((size_t)(vi - 1) * (uj - 1)) {
    for (size_t k = 0; k < m_.size(); k++) m_[k] = mask_rho[k] == 1;
  }

  // 0 on success
  int build() {
    prune_mask();
    count_points();
    if (bnum_ == 0) return -1;
    classify();
    if (!resolve_crosses()) return -1;
    if (!walk_water()) return -1;
    if (!walk_islands()) return -1;
    return 0;
  }

  const std::vector<Point>& points() const { return pts_; }
  bool water(int i, int j) const { return m_[(size_t)(i - 1) + (size_t)vi_ * (j - 1)] != 0; }

 private:
  int vi_, uj_, bnum_ = 0;
  std::vector<char> m_;
  std::vector<unsigned char> form_, visits_;   // visits: 0, 1 (a cross passed once), 2 = used up
  std::vector<Point> pts_;
  size_t poly_first_ = 0;   // index in pts_ of the first point of the polygon being walked
  int poly_ = 1;
  bool poly_closed_ = false, all_used_ = false;

  unsigned char& form(int i, int j) { return form_[(size_t)(i - 1) + (size_t)(vi_ - 1) * (j - 1)]; }
  unsigned char& visits(int i, int j) { return visits_[(size_t)(i - 1) + (size_t)(vi_ - 1) * (j - 1)]; }
  char& mask(int i, int j) { return m_[(size_t)(i - 1) + (size_t)vi_ * (j - 1)]; }
  static bool cross(int f) { return f == 6 || f == 9; }

  // :188-218 a water node with fewer than two water neighbours is not covered by the u / v grids
  void prune_mask() {
    for (bool changed = true; changed;) {
      changed = false;
      for (int j = 1; j <= uj_; j++)
        for (int i = 1; i <= vi_; i++) {
          if (!mask(i, j)) continue;
          const int nb = (i > 1 && mask(i - 1, j)) + (i < vi_ && mask(i + 1, j)) + (j > 1 && mask(i, j - 1)) + (j < uj_ && mask(i, j + 1));
          if (nb < 2) { mask(i, j) = 0; changed = true; }
        }
    }
  }

  // :229-260 one point per pair of interior neighbours of different mask, one per water node next to the rim
  void count_points() {
    bnum_ = 0;
    for (int i = 2; i <= vi_ - 1; i++)
      for (int j = 2; j <= uj_ - 1; j++) {
        if (i < vi_ - 1 && mask(i, j) != mask(i + 1, j)) bnum_++;
        if (j < uj_ - 1 && mask(i, j) != mask(i, j + 1)) bnum_++;
      }
    for (int i = 2; i <= vi_ - 1; i++) bnum_ += (mask(i, 2) != 0) + (mask(i, uj_ - 1) != 0);
    for (int j = 2; j <= uj_ - 1; j++) bnum_ += (mask(2, j) != 0) + (mask(vi_ - 1, j) != 0);
    pts_.reserve((size_t)bnum_);
  }

  void classify() {
    visits_.assign(form_.size(), 0);
    for (int i = 1; i <= vi_ - 1; i++)
      for (int j = 1; j <= uj_ - 1; j++) {
        const int f = (!mask(i, j + 1)) * 1 + (!mask(i, j)) * 2 + (!mask(i + 1, j + 1)) * 4 + (!mask(i + 1, j)) * 8;
        form(i, j) = (unsigned char)f;
        if (f == 0 || f == 15) visits(i, j) = 2;
      }
  }

  // ---- moving from element to element
  struct Cursor { int i, j, dir; bool rim; };

  // which side a walk round water leaves an interior element by
  int exit_side_water(int f, int dir) const {
    if (f < 16) return EXIT_WATER[f];
    if (f <= 17) return ((f == 16 && dir == 2) || (f == 17 && dir == 8)) ? RIGHT : LEFT;
    return ((f == 18 && dir == 4) || (f == 19 && dir == 6)) ? TOP : BOTTOM;
  }
  int exit_side_island(int f, int dir) const {
    if (f < 16) return EXIT_ISLAND[f];
    if (f <= 17) return ((f == 16 && dir == 4) || (f == 17 && dir == 6)) ? TOP : BOTTOM;
    return ((f == 18 && dir == 2) || (f == 19 && dir == 8)) ? RIGHT : LEFT;
  }
  // forms through which a path running along the rim strip in direction dir stays on it
  static bool stays_on_rim(int f, int dir) {
    switch (dir) {
      case 2: return f == 0 || f == 1 || f == 4 || f == 5;
      case 4: return f == 0 || f == 1 || f == 2 || f == 3;
      case 6: return f == 0 || f == 4 || f == 8 || f == 12;
      default: return f == 0 || f == 2 || f == 8 || f == 10;
    }
  }

  // getNext (:1283-1513): the probe used to resolve crosses.  No points are emitted.
  void probe_step(Cursor& c, int f) {
    if (c.rim) {
      if (!stays_on_rim(f, c.dir)) { c.rim = false; return; }
      switch (c.dir) {   // (turns the corner when it reaches the element BEFORE the corner one)
        case 2: c.i--; if (c.i == 2) { c.dir = 4; c.j--; } break;
        case 4: c.j--; if (c.j == 2) { c.dir = 8; c.i++; } break;
        case 6: c.j++; if (c.j == uj_ - 2) { c.dir = 2; c.i--; } break;
        default: c.i++; if (c.i == vi_ - 2) { c.dir = 6; c.j++; } break;
      }
      return;
    }
    marking_ = false;
    leave_interior(c, exit_side_water(f, c.dir), (void (*)(bool, int, int)) nullptr);
  }

  // leaving an interior element through `side`: into the neighbour, or -- from the elements next to the rim
  // strip -- onto the strip, diagonally, turning at once in a corner.  emit = the walk's add() calls.
  template <typename Emit>
  void leave_interior(Cursor& c, int side, Emit* emit) {
    auto put = [&](bool onU, int i, int j) { if (emit) (*emit)(onU, i, j); };
    if (side < 0) return;   // an all-water / all-land element: getNext does nothing either
    switch (side) {
      case BOTTOM:
        put(false, c.i, c.j);
        if (c.i == 2) {
          c.rim = true; c.i--;
          put(true, c.i, c.j); used_up(c);
          c.j--;
          if (c.j == 1) { put(false, c.i + 1, c.j); used_up(c); c.i++; c.dir = 8; } else c.dir = 4;
        } else { c.i--; c.dir = 2; }
        break;
      case LEFT:
        put(true, c.i, c.j);
        if (c.j == 2) {
          c.rim = true; c.j--;
          put(false, c.i + 1, c.j); used_up(c);
          c.i++;
          if (c.i == vi_ - 1) { put(true, c.i, c.j + 1); used_up(c); c.j++; c.dir = 6; } else c.dir = 8;
        } else { c.j--; c.dir = 4; }
        break;
      case RIGHT:
        put(true, c.i, c.j + 1);
        if (c.j == uj_ - 2) {
          c.rim = true; c.j++;
          // (the reference adds U(ipos, jpos) here, :764-768 -- the point it has just added, not the v point
          //  of the corner; the three other sides add the other kind.  Kept: the arrays must be the reference's.)
          put(true, c.i, c.j); used_up(c);
          c.i--;
          if (c.i == 1) { put(false, c.i, c.j); used_up(c); c.j--; c.dir = 4; } else c.dir = 2;
        } else { c.j++; c.dir = 6; }
        break;
      default:  // TOP
        put(false, c.i + 1, c.j);
        if (c.i == vi_ - 2) {
          c.rim = true; c.i++;
          put(true, c.i, c.j + 1); used_up(c);
          c.j++;
          if (c.j == uj_ - 1) { put(false, c.i, c.j); used_up(c); c.i--; c.dir = 2; } else c.dir = 6;
        } else { c.i++; c.dir = 8; }
        break;
    }
  }
  // (the walk marks the rim elements it passes diagonally as used; the probe has nothing to mark)
  bool marking_ = false;
  void used_up(const Cursor& c) {
    if (marking_) visits(c.i, c.j) = 2;
  }

  // ---- crosses (:333-614)
  bool resolve_crosses() {
    int open = 0;
    for (unsigned char f : form_) open += cross(f);
    int before = -1;
    const long cap = 8L * vi_ * uj_ + 64;
    while (open > 0) {
      if (before == open) {   // nothing was resolved in a whole sweep: the defaults
        for (auto& f : form_) { if (f == 6) f = 16; else if (f == 9) f = 19; }
        break;
      }
      before = open;
      for (int i = 1; i <= vi_ - 1 && open > 0; i++)
        for (int j = 1; j <= uj_ - 1 && open > 0; j++) {
          const int f = form(i, j);
          if (!cross(f)) continue;
          if (i == 1 || j == 1 || i == vi_ - 1 || j == uj_ - 1) {   // on the rim strip
            form(i, j) = (f == 6) ? 17 : 18;
            open--;
            continue;
          }
          // two probes leave the cross by its two possible exits; the one that comes back decides
          Cursor a{i, j, 5, false}, b{i, j, 5, false};
          if (f == 6) {   // exits left and right
            if (j == 2) { a.rim = true; a.j--; a.i++; a.dir = 8; if (a.i == vi_ - 1) { a.j++; a.dir = 6; } }
            else { a.j--; a.dir = 4; }
            if (j == uj_ - 2) { b.rim = true; b.j++; b.i--; b.dir = 2; if (b.i == 1) { b.j--; b.dir = 4; } }
            else { b.j++; b.dir = 6; }
          } else {        // exits up and down
            if (i == vi_ - 2) { a.rim = true; a.i++; a.j++; a.dir = 6; if (a.j == uj_ - 1) { a.i--; a.dir = 2; } }
            else { a.i++; a.dir = 8; }
            if (i == 2) { b.rim = true; b.i--; b.j--; if (b.j == 1) { b.i++; b.dir = 8; } else b.dir = 4; }
            else { b.i--; b.dir = 2; }
          }
          bool stuck_a = false, stuck_b = false;
          for (long step = 0;; step++) {
            if (step > cap) return false;
            if (a.i == i && a.j == j) {
              if (f == 6) { if (a.dir == 2) form(i, j) = 16; else if (a.dir == 8) form(i, j) = 17; }
              else { if (a.dir == 4) form(i, j) = 19; else if (a.dir == 6) form(i, j) = 18; }
              open--;
              break;
            }
            if (b.i == i && b.j == j) {
              if (f == 6) { if (b.dir == 2) form(i, j) = 17; else if (b.dir == 8) form(i, j) = 16; }
              else { if (b.dir == 4) form(i, j) = 18; else if (b.dir == 6) form(i, j) = 19; }
              open--;
              break;
            }
            stuck_a = stuck_a || cross(form(a.i, a.j));
            stuck_b = stuck_b || cross(form(b.i, b.j));
            if (stuck_a && stuck_b) break;   // both ran into an unresolved cross: later
            if (!stuck_a) probe_step(a, form(a.i, a.j));
            if (!stuck_b) probe_step(b, form(b.i, b.j));
          }
        }
    }
    return true;
  }

  // ---- add (:1223-1280)
  void emit(bool onU, int i, int j) {
    if (poly_closed_) return;   // (the walk's `if(.NOT. polydone) CALL add`)
    if (pts_.size() > poly_first_ && pts_[poly_first_].onU == onU && pts_[poly_first_].i == i && pts_[poly_first_].j == j) {
      poly_closed_ = true;
      poly_++;
      poly_first_ = pts_.size();
      if ((int)pts_.size() == bnum_) all_used_ = true;
      return;
    }
    if ((int)pts_.size() >= bnum_) { poly_closed_ = true; all_used_ = true; return; }
    pts_.push_back(Point{onU, i, j, poly_});
  }
  void pass(const Cursor& c) {   // :665-676 a cross is used up at its second passage
    unsigned char& v = visits(c.i, c.j);
    if (form(c.i, c.j) > 15 && v == 0) v = 1; else v = 2;
  }

  // ---- the walk round the water (:623-900)
  bool walk_water() {
    Cursor c{0, 0, 5, false};
    if (mask(2, 2)) { c.i = 1; c.j = 1; }
    else {
      for (int i = 2; i <= vi_ - 2 && !c.i; i++)
        for (int j = 2; j <= uj_ - 2; j++)
          if (form(i, j) > 0 && form(i, j) < 15) { c.i = i; c.j = j; break; }
      if (!c.i) return false;
    }
    auto put = [this](bool onU, int i, int j) { emit(onU, i, j); };
    if (c.i == 1 && c.j == 1) {   // starts in the bottom-left corner of the rim strip
      // (the very first point: add's test does not see it as the closing one, nothing is stored yet)
      emit(false, 2, 1);
      c.rim = true; c.dir = 8; c.i = 2;
    }
    const long cap = 8L * vi_ * uj_ + 64;
    for (long step = 0; !poly_closed_; step++) {
      if (step > cap) return false;
      pass(c);
      const int f = form(c.i, c.j);
      if (c.rim) {
        if (!stays_on_rim(f, c.dir)) { c.rim = false; continue; }
        switch (c.dir) {   // one element along the strip; in the corner element, turn
          case 2:
            put(false, c.i, c.j); c.i--;
            if (c.i == 1) { put(true, c.i, c.j); visits(c.i, c.j) = 2; c.dir = 4; c.j--; }
            break;
          case 4:
            put(true, c.i, c.j); c.j--;
            if (c.j == 1) { put(false, c.i + 1, c.j); visits(c.i, c.j) = 2; c.dir = 8; c.i++; }
            break;
          case 6:
            put(true, c.i, c.j + 1); c.j++;
            if (c.j == uj_ - 1) { put(false, c.i, c.j); visits(c.i, c.j) = 2; c.dir = 2; c.i--; }
            break;
          default:
            put(false, c.i + 1, c.j); c.i++;
            if (c.i == vi_ - 1) { put(true, c.i, c.j + 1); visits(c.i, c.j) = 2; c.dir = 6; c.j++; }
            break;
        }
        continue;
      }
      const int side = exit_side_water(f, c.dir);
      if (side == NONE) return false;
      marking_ = true;
      leave_interior(c, side, &put);
    }
    marking_ = false;
    last_dir_ = c.dir;
    return true;
  }

  // ---- the islands (:904-1013): first unused boundary element in (i, j) order, round the land
  bool walk_islands() {
    const long cap = 8L * vi_ * uj_ + 64;
    for (long n = 0; !all_used_; n++) {
      if (n > (long)vi_ * uj_) return false;
      Cursor c{0, 0, 5, false};
      for (int i = 2; i <= vi_ - 2 && !c.i; i++)
        for (int j = 2; j <= uj_ - 2; j++)
          if (form(i, j) < 15 && visits(i, j) != 2) { c.i = i; c.j = j; break; }
      if (!c.i) return false;
      poly_closed_ = false;
      // (the walk's direction variable is NOT reset between polygons in the reference either)
      c.dir = last_dir_;
      for (long step = 0; !poly_closed_; step++) {
        if (step > cap) return false;
        pass(c);
        const int side = exit_side_island(form(c.i, c.j), c.dir);
        switch (side) {
          case RIGHT: emit(true, c.i, c.j + 1); c.j++; break;
          case BOTTOM: emit(false, c.i, c.j); c.i--; break;
          case TOP: emit(false, c.i + 1, c.j); c.i++; break;
          case LEFT: emit(true, c.i, c.j); c.j--; break;
          default: return false;
        }
        c.dir = DIR_OF_SIDE[side];
        if (c.i < 1 || c.i > vi_ - 1 || c.j < 1 || c.j > uj_ - 1) return false;
      }
      last_dir_ = c.dir;
    }
    return true;
  }
  int last_dir_ = 5;
};

// createBounds' output arrays (boundary_module.f90:1040-1215) from the walk: bx / by (the main polygon, closed),
// hx / hy / hid (the islands, each closed, ids from 1001), bnd_x / bnd_y / land (the segments; land = 0 between two
// open-ocean points).  x_u(vi-1,uj), x_v(vi,uj-1) column-major; every output has room for `cap` entries (2*cap for
// bnd_x / bnd_y).  Returns 0, or 1 when the mask cannot be walked or the arrays are too small.
inline int create_bounds_arrays(int32_t vi, int32_t uj, const int32_t* mask_rho, const double* x_u, const double* y_u,
                                const double* x_v, const double* y_v, int32_t cap, int32_t* nbounds_o, int32_t* maxbound_o,
                                int32_t* maxisland_o, int32_t* numislands_o, double* bx, double* by, double* hx, double* hy,
                                int32_t* hid, double* bnd_x, double* bnd_y, int32_t* land) {
  Bounds B(vi, uj, mask_rho);
  if (B.build() != 0) return 1;
  const std::vector<Point>& P = B.points();
  const int ui = vi - 1, numpoly = P.back().poly, numislands = numpoly - 1;
  std::vector<int> size(numpoly + 1, 0), first(numpoly + 2, 0);
  for (const Point& p : P) size[p.poly]++;
  for (int m = 1; m <= numpoly; m++) first[m + 1] = first[m] + size[m];
  const int maxbound = size[1] + 1;
  int maxisland = numislands;
  for (int m = 2; m <= numpoly; m++) maxisland += size[m];
  const int nbounds = maxbound + maxisland - numislands - 1;
  if (maxbound > cap || maxisland > cap || nbounds > cap) return 1;
  auto xy = [&](const Point& p, double& x, double& y) {
    if (p.onU) { x = x_u[(size_t)(p.i - 1) + (size_t)ui * (p.j - 1)]; y = y_u[(size_t)(p.i - 1) + (size_t)ui * (p.j - 1)]; }
    else { x = x_v[(size_t)(p.i - 1) + (size_t)vi * (p.j - 1)]; y = y_v[(size_t)(p.i - 1) + (size_t)vi * (p.j - 1)]; }
  };
  // open ocean: a point on the rim of the grid next to a water node of the outermost ring (:1071-1079)
  auto open_ocean = [&](const Point& p) {
    if (p.onU) return (p.i == 1 && B.water(1, p.j)) || (p.i == vi - 1 && B.water(vi, p.j));
    return (p.j == 1 && B.water(p.i, 1)) || (p.j == uj - 1 && B.water(p.i, uj));
  };
  std::vector<char> oo(maxbound, 0);
  for (int k = 0; k < maxbound; k++) {   // the main polygon, closed by repeating its first point
    const Point& p = P[first[1] + (k < size[1] ? k : 0)];
    xy(p, bx[k], by[k]);
    oo[k] = open_ocean(p);
  }
  int c = 0;
  for (int m = 2; m <= numpoly; m++)
    for (int k = 0; k <= size[m]; k++, c++) {
      xy(P[first[m] + (k < size[m] ? k : 0)], hx[c], hy[c]);
      hid[c] = m + 999;   // island ids start at 1001
    }
  for (int i = 0; i < maxbound - 1; i++) {
    bnd_x[2 * i] = bx[i]; bnd_y[2 * i] = by[i]; bnd_x[2 * i + 1] = bx[i + 1]; bnd_y[2 * i + 1] = by[i + 1];
    land[i] = (oo[i] && oo[i + 1]) ? 0 : 1;
  }
  int q = maxbound - 1;
  for (int i = 0; i + 1 < maxisland; i++) {
    if (hid[i] != hid[i + 1]) continue;   // (the closing point of one island and the first of the next)
    bnd_x[2 * q] = hx[i]; bnd_y[2 * q] = hy[i]; bnd_x[2 * q + 1] = hx[i + 1]; bnd_y[2 * q + 1] = hy[i + 1];
    land[q] = 1;
    q++;
  }
  *nbounds_o = nbounds; *maxbound_o = maxbound; *maxisland_o = maxisland; *numislands_o = numislands;
  return q == nbounds ? 0 : 1;
}

}  // namespace ltb
