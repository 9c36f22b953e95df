// Event-driven topology update of the grain graph, HOST code (no kernel in this file): ggnn_topology_update
// (include/ggnn.h) restates `GrainNN_classifier.update` of the reference (models.py:612-842 with its helpers
// `delete_grain_index` :861-893, `switching_edge_index` :896-1051, `point_in_triangle` :1055-1070, `periodic_move`
// :1103-1106) for the periodic, nucleation-free configuration every shipped script runs (test.py:88).
//
// The reference answers every lookup ("the columns whose source is p", "the junctions of grain g", "grains left with two
// junctions") with a mask over a WHOLE edge list; round 5 measured 57-75 ms of host time per eventful step at the 10k-grain
// graph that way (19-23 grains vanish), 8-9 ms with the lookups answered from column indices in Python.  Here they are
// answered from the same indices in native code: columns grouped by value once per call (counting sort), later rewrites in a
// side table, every answer filtered by the list's CURRENT content and returned in increasing column order -- what
// `(row == v).nonzero()` gives -- plus a running count of columns per grain.
//
// Bit-exactness contract (tests/golden/golden_cfg1_events*.npz, produced by the unmodified reference; oracle/topology_scan.py
// on large random event sequences): identical edge-list COLUMN ORDER (edges are rewritten in place, new edges appended, dead
// columns dropped at the end, never re-sorted), identical masks, identical fp32 junction coordinates (every fp32 operation
// below is a single rounded operation in the reference's order: contraction is off).  Reference behaviours that look
// accidental are kept because the trained models were run with them, each marked KEEP.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "ggnn.h"

#pragma clang fp contract(off)

namespace {

constexpr int64_t DEAD = -1;            // marker of a removed column until the final clean-up
constexpr float JOINT_SCALING = 5.0f;   // models.py:398 scaling['joint']

struct Refused {};   // thrown with the message already in args.error

// `cols(v)`: the columns c with row[c] == v, in increasing order, without scanning the row.  Exact at every moment: a
// column is moved between the lists of its old and its new value when it is rewritten (round 5 rebuilt this index by a counting
// sort at every call and filtered stale entries at every lookup: four O(E) passes per call for a handful of events).
struct KeyIndex {
  std::vector<std::vector<int64_t>> at;
  void init(int64_t n_keys) { at.assign((size_t)n_keys, {}); }
  void add(int64_t key, int64_t col) {
    auto& v = at[(size_t)key];
    if (v.empty() || v.back() < col) v.push_back(col);
    else v.insert(std::lower_bound(v.begin(), v.end(), col), col);
  }
  void drop(int64_t key, int64_t col) {
    auto& v = at[(size_t)key];
    auto it = std::lower_bound(v.begin(), v.end(), col);
    if (it != v.end() && *it == col) v.erase(it);
  }
  void cols(int64_t v, std::vector<int64_t>& out) const {
    if (v >= 0 && v < (int64_t)at.size()) out = at[(size_t)v];
    else out.clear();
  }
};

}  // namespace

// The lists of one trajectory between two calls (ggnn_topology_open / _apply / _export / _close, include/ggnn.h): columns keep a
// STABLE number for the session's life -- rewritten in place, new ones appended, removed ones left behind as DEAD -- so the
// indices and the per-grain counts persist and a call costs what its events touch, plus one pass that renumbers the live
// columns for the caller (the reference drops dead columns at the end of every update: the caller's column c is the c-th live
// stable column).  Every write of a call is journalled: a refused call leaves the session and the caller's arrays as they were.
struct ggnn_topology_session {
  int64_t n_joint = 0, n_grain = 0;
  std::vector<int64_t> arr[4];          // pp0, pp1, pq0, pq1 by stable column
  KeyIndex idx[4];
  // The caller numbers the LIVE columns 0 .. n - 1 in stable order.  Only the junction list is ever addressed by the
  // caller (edge_prob[c]): a Fenwick tree over the liveness of its stable columns answers "the c-th live column" in
  // O(log E) and follows a removed / appended column in O(log E) -- no per-call renumbering pass.
  std::vector<int32_t> fen;             // 1-based partial sums of live_pp
  int64_t fen_n = 0, n_live[2] = {0, 0};   // tree size (a power of two >= the stable columns it can hold); live pp / pq columns
  std::vector<int64_t> n_of_grain;      // columns per grain: the counts of np.unique(pq[1])
  std::vector<int64_t> few;             // grains with one or two columns, unsorted (membership checked against n_of_grain)
  struct Undo { int kind; int64_t col, old; };          // kind 0..3: arr[kind][col] was `old`; 4: two pp columns appended
  struct UndoF { float* p; float old; };
  struct UndoM { int64_t* p; int64_t old; };
  std::vector<Undo> undo;
  std::vector<UndoF> undo_f;
  std::vector<UndoM> undo_m;

  void recount(int64_t g, int by) {
    const int64_t n = n_of_grain[(size_t)g] += by;
    if (n >= 1 && n <= 2) few.push_back(g);   // (duplicates and stale entries are dropped when the list is read)
  }
  void fen_add(int64_t col, int by) {
    for (int64_t i = col + 1; i <= fen_n; i += i & -i) fen[(size_t)i] += by;
  }
  int64_t kth_live_pp(int64_t c) const {   // stable column of the caller's column c (0 <= c < n_live[0])
    int64_t pos = 0, rest = c + 1;
    for (int64_t step = fen_n; step > 0; step >>= 1)
      if (pos + step <= fen_n && fen[(size_t)(pos + step)] < rest) pos += step, rest -= fen[(size_t)pos];
    return pos;   // (0-based stable column: the first position whose prefix count reaches c + 1)
  }
  void raw_put(int kind, int64_t col, int64_t v) {
    int64_t& slot = arr[kind][(size_t)col];
    const int64_t old = slot;
    if (old == v) return;
    if ((kind & 1) == 0 && (old == -1 || v == -1)) {   // row 0 of a list decides whether a column is live
      const int by = v == -1 ? -1 : +1;
      n_live[kind >> 1] += by;
      if (kind == 0) fen_add(col, by);
    }
    if (old != -1) {
      idx[kind].drop(old, col);
      if (kind == 3) recount(old, -1);
    }
    slot = v;
    if (v != -1) {
      idx[kind].add(v, col);
      if (kind == 3) recount(v, +1);
    }
  }
  void put(int kind, int64_t col, int64_t v) {
    const int64_t old = arr[kind][(size_t)col];
    if (old == v) return;
    undo.push_back({kind, col, old});
    raw_put(kind, col, v);
  }
  int64_t append_pp(int64_t a, int64_t b) {
    const int64_t col = (int64_t)arr[0].size();
    if (col >= fen_n) throw std::length_error("the junction edge list outgrew the session's room (2 columns per grain)");
    arr[0].push_back(-1), arr[1].push_back(-1);
    undo.push_back({4, col, 0});
    raw_put(0, col, a), raw_put(1, col, b);
    return col;
  }
  void put_f(float* p, float v) {
    undo_f.push_back({p, *p});
    *p = v;
  }
  void put_m(int64_t* p, int64_t v) {
    undo_m.push_back({p, *p});
    *p = v;
  }
  void commit() { undo.clear(), undo_f.clear(), undo_m.clear(); }
  void rollback() {
    for (size_t k = undo.size(); k-- > 0;) {
      const Undo& u = undo[k];
      if (u.kind < 4) raw_put(u.kind, u.col, u.old);
      else raw_put(0, u.col, -1), raw_put(1, u.col, -1), arr[0].pop_back(), arr[1].pop_back();
    }
    for (size_t k = undo_f.size(); k-- > 0;) *undo_f[k].p = undo_f[k].old;
    for (size_t k = undo_m.size(); k-- > 0;) *undo_m[k].p = undo_m[k].old;
    commit();
  }
  // when most stable columns are dead: a fresh start from the live ones (keeps the O(E) passes of export short)
  void rebase_if_sparse() {
    if ((int64_t)arr[0].size() <= 2 * n_live[0] + 1024 && (int64_t)arr[2].size() <= 2 * n_live[1] + 1024) return;
    std::vector<int64_t> live[4];
    for (int k = 0; k < 4; ++k)
      for (size_t c = 0; c < arr[k].size(); ++c)
        if (arr[k & 2][c] != -1) live[k].push_back(arr[k][c]);
    load(live[0].data(), live[1].data(), (int64_t)live[0].size(), live[2].data(), live[3].data(), (int64_t)live[2].size());
  }
  void load(const int64_t* p0, const int64_t* p1, int64_t n_pp, const int64_t* q0, const int64_t* q1, int64_t n_pq) {
    const int64_t* src[4] = {p0, p1, q0, q1};
    for (int k = 0; k < 4; ++k) {
      const int64_t n = k < 2 ? n_pp : n_pq;
      arr[k].assign(src[k], src[k] + n);
      idx[k].init(k == 3 ? n_grain : n_joint);
      for (int64_t c = 0; c < n; ++c) idx[k].at[(size_t)arr[k][(size_t)c]].push_back(c);
    }
    n_of_grain.assign((size_t)n_grain, 0);
    for (int64_t c = 0; c < n_pq; ++c) ++n_of_grain[(size_t)arr[3][(size_t)c]];
    few.clear();
    for (int64_t g = 0; g < n_grain; ++g)
      if (n_of_grain[(size_t)g] >= 1 && n_of_grain[(size_t)g] <= 2) few.push_back(g);
    n_live[0] = n_pp, n_live[1] = n_pq;
    fen_n = 1;
    while (fen_n < n_pp + 2 * n_grain + 2) fen_n <<= 1;   // (a removed grain appends two columns)
    fen.assign((size_t)fen_n + 1, 0);
    for (int64_t i = 1; i <= fen_n; ++i) {   // O(n) construction: every node passes its sum on to its parent
      if (i <= n_pp) fen[(size_t)i] += 1;
      const int64_t up = i + (i & -i);
      if (up <= fen_n) fen[(size_t)up] += fen[(size_t)i];
    }
  }
};

namespace {

using Session = ggnn_topology_session;

// One call of the reference's update on a session's lists.
struct Topology {
  ggnn_topology_args& A;
  Session& S;
  std::vector<int64_t>&pp0, &pp1, &pq0, &pq1;
  KeyIndex &ipp0, &ipp1, &ipq0, &ipq1;
  std::vector<int64_t>& n_of_grain;
  std::vector<int64_t>& few;
  int64_t n_dead = 0;                // pq columns killed by THIS call (the reference's list still holds them as -1)

  [[noreturn]] void refuse(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(A.error, sizeof(A.error), fmt, ap);
    va_end(ap);
    throw Refused{};
  }
  int64_t joint(int64_t j) {
    if (j < 0 || j >= A.n_joint) refuse("junction index %lld outside [0, %lld)", (long long)j, (long long)A.n_joint);
    return j;
  }
  int64_t grain_id(int64_t g) {
    if (g < 0 || g >= A.n_grain) refuse("grain index %lld outside [0, %lld)", (long long)g, (long long)A.n_grain);
    return g;
  }
  float* xj(int64_t j) { return A.x_joint + joint(j) * A.ldx; }
  float* yj(int64_t j) { return A.y_joint + joint(j) * 2; }
  bool is_active(int64_t j) { return A.active_joint == nullptr || A.active_joint[joint(j)] != 0; }

  Topology(ggnn_topology_args& a, Session& s)
      : A(a), S(s), pp0(s.arr[0]), pp1(s.arr[1]), pq0(s.arr[2]), pq1(s.arr[3]), ipp0(s.idx[0]), ipp1(s.idx[1]),
        ipq0(s.idx[2]), ipq1(s.idx[3]), n_of_grain(s.n_of_grain), few(s.few) {}

  // -- writes that the indices, the per-grain counts and the journal follow --
  void set_grain(int64_t col, int64_t g) { S.put(3, col, grain_id(g)); }
  void kill_pq(const std::vector<int64_t>& cols) {
    for (int64_t c : cols) {
      ++n_dead;
      S.put(2, c, DEAD), S.put(3, c, DEAD);
    }
  }
  void kill_pp(const std::vector<int64_t>& cols) {
    for (int64_t c : cols) S.put(0, c, DEAD), S.put(1, c, DEAD);
  }
  bool has_pq(int64_t j, int64_t g) {
    std::vector<int64_t> c;
    ipq0.cols(j, c);
    for (int64_t k : c)
      if (pq1[k] == g) return true;
    return false;
  }

  // -- models.py:861-893: a grain reduced to two junctions disappears: its junctions p1, p2 die and their two outer
  // neighbours are joined by a new edge pair appended at the end --
  void remove_two_sided_grain(int64_t grain) {
    if (grain == DEAD) refuse("the dead-column marker was counted as a two-sided grain");
    std::vector<int64_t> corners, c;
    ipq1.cols(grain, corners);
    if (corners.size() != 2) refuse("grain %lld has %zu junctions, expected 2", (long long)grain, corners.size());
    const int64_t p1 = pq0[corners[0]], p2 = pq0[corners[1]];
    int64_t n1 = DEAD, n2 = DEAD;
    ipp0.cols(p1, c);
    for (int64_t k : c)
      if (pp1[k] != p2) { n1 = pp1[k]; break; }
    ipp0.cols(p2, c);
    for (int64_t k : c)
      if (pp1[k] != p1) { n2 = pp1[k]; break; }
    if (n1 == DEAD || n2 == DEAD)
      refuse("junctions %lld, %lld of grain %lld have no outer neighbour", (long long)p1, (long long)p2, (long long)grain);
    S.append_pp(n1, n2), S.append_pp(n2, n1);
    S.put_m(&A.mask_grain[grain_id(grain)], 0);
    S.put_m(&A.mask_joint[joint(p1)], 0);
    S.put_m(&A.mask_joint[joint(p2)], 0);
    ipq1.cols(grain, c);
    kill_pq(c);
    for (int64_t j : {p1, p2}) {
      ipq0.cols(j, c);
      kill_pq(c);
      ipp0.cols(j, c);
      kill_pp(c);
      ipp1.cols(j, c);
      kill_pp(c);
    }
  }

  // models.py:708-717 / 741-750.  KEEP: the DEAD marker itself takes part in the count (it never has <= 2 columns once a
  // grain has been removed).  np.unique order: ascending.
  std::vector<int64_t> remove_all_two_sided() {
    std::sort(few.begin(), few.end());
    few.erase(std::unique(few.begin(), few.end()), few.end());
    few.erase(std::remove_if(few.begin(), few.end(), [&](int64_t g) { return n_of_grain[g] < 1 || n_of_grain[g] > 2; }),
              few.end());
    std::vector<int64_t> found;
    if (n_dead >= 1 && n_dead <= 2) found.push_back(DEAD);
    found.insert(found.end(), few.begin(), few.end());
    for (int64_t g : found) remove_two_sided_grain(g);
    return found;
  }

  // Periodic image of p nearest to ref (models.py:1103-1106), fp32
  static void wrap_to(const float* p, const float* ref, float* out) {
    for (int d = 0; d < 2; ++d) {
      const float rel = p[d] - ref[d];
      out[d] = (p[d] - (rel > 0.5f ? 1.0f : 0.0f)) + (rel < -0.5f ? 1.0f : 0.0f);
    }
  }
  // models.py:1055-1070, evaluated in fp32 in the reference's operation order
  static bool inside_triangle(const float* t, const float* v1, const float* v2, const float* v3) {
    float a[2], b[2], c[2];
    wrap_to(v1, t, a), wrap_to(v2, t, b), wrap_to(v3, t, c);
    auto side = [](const float* p, const float* q, const float* r) {
      const float u = (p[0] - r[0]) * (q[1] - r[1]);
      const float v = (q[0] - r[0]) * (p[1] - r[1]);
      return u - v;
    };
    const float d[3] = {side(t, a, b), side(t, b, c), side(t, c, a)};
    const bool neg = d[0] < 0 || d[1] < 0 || d[2] < 0, pos = d[0] > 0 || d[1] > 0 || d[2] > 0;
    return !(neg && pos);
  }

  // -- models.py:896-1051: neighbour switching (T1) of the junction-junction columns `cols`, in order.  Returns grains that
  // turn out to be squeezed between the switching junctions (forced eliminations). --
  std::vector<int64_t> switch_edges(const std::vector<int64_t>& cols, bool vanishing, int64_t vanishing_grain) {
    std::vector<int64_t> forced, touched;
    for (int64_t c : cols) touched.push_back(pp0[c]), touched.push_back(pp1[c]);
    std::sort(touched.begin(), touched.end());
    touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
    for (int64_t p : touched) {   // back to the position before this step
      float *x = xj(p), *y = yj(p);
      S.put_f(&x[0], x[0] - y[0] / JOINT_SCALING);
      S.put_f(&x[1], x[1] - y[1] / JOINT_SCALING);
    }
    std::vector<int64_t> c1, c2, e1, e2, tmp;
    for (size_t k = 0; k < cols.size(); ++k) {
      const int64_t p1 = pp0[cols[k]], p2 = pp1[cols[k]];
      if (!(is_active(p1) && is_active(p2))) continue;
      ipq0.cols(p1, c1), ipq0.cols(p2, c2);
      std::vector<int64_t> g1, g2, n1, n2;
      for (int64_t c : c1) g1.push_back(pq1[c]);
      for (int64_t c : c2) g2.push_back(pq1[c]);
      ipp0.cols(p1, tmp);
      e1.clear();
      for (int64_t c : tmp)
        if (pp1[c] != p2) e1.push_back(c);   // columns p1 -> its other neighbours
      ipp0.cols(p2, tmp);
      e2.clear();
      for (int64_t c : tmp)
        if (pp1[c] != p1) e2.push_back(c);
      for (int64_t c : e1) n1.push_back(pp1[c]);
      for (int64_t c : e2) n2.push_back(pp1[c]);
      auto in = [](const std::vector<int64_t>& v, int64_t x) { return std::find(v.begin(), v.end(), x) != v.end(); };
      std::vector<int64_t> grow1, grow2, shrink;   // grains of p1 only / of p2 only / of both
      for (int64_t g : g1) (in(g2, g) ? shrink : grow1).push_back(g);
      for (int64_t g : g2)
        if (!in(g1, g)) grow2.push_back(g);
      if (shrink.size() != 2 || grow1.size() != 1 || grow2.size() != 1)
        refuse("junctions %lld, %lld do not share exactly two grains", (long long)p1, (long long)p2);
      if (c1.size() < 3 || c2.size() < 3 || n1.size() < 2 || n2.size() < 2)
        refuse("junctions %lld, %lld do not have three grains and three neighbours each", (long long)p1, (long long)p2);
      // (models.py:951-965 goes on with the FIRST TWO other neighbours of a junction that has more)
      e1.resize(2), n1.resize(2), e2.resize(2), n2.resize(2);
      const int64_t sa = shrink[0], sb = shrink[1];
      std::vector<int64_t> d1, d2;   // each junction's columns of (sa, sb)
      for (int i = 0; i < 3; ++i)
        if (g1[i] == sa) d1.push_back(c1[i]);
      for (int i = 0; i < 3; ++i)
        if (g1[i] == sb) d1.push_back(c1[i]);
      for (int i = 0; i < 3; ++i)
        if (g2[i] == sa) d2.push_back(c2[i]);
      for (int i = 0; i < 3; ++i)
        if (g2[i] == sb) d2.push_back(c2[i]);
      if (d1.size() < 2 || d2.size() < 2) refuse("junctions %lld, %lld: inconsistent grain columns", (long long)p1, (long long)p2);
      // order each junction's two outer neighbours as (the one on grain sa, the one on sb)
      if (!has_pq(n1[0], sa)) std::reverse(e1.begin(), e1.end()), std::reverse(n1.begin(), n1.end());
      if (!has_pq(n2[0], sa)) std::reverse(e2.begin(), e2.end()), std::reverse(n2.begin(), n2.end());
      int64_t a1 = n1[0], b1 = n1[1], a2 = n2[0], b2 = n2[1];
      if (!vanishing && (a1 == a2 || b1 == b2)) continue;   // a triangle would collapse: not a pure switch
      if (a1 == a2 && !(vanishing && sa == vanishing_grain)) forced.push_back(sa);
      if (b1 == b2 && !(vanishing && sb == vanishing_grain)) forced.push_back(sb);
      // both junctions move to the (periodic) mid point of the edge
      float *x1 = xj(p1), *x2 = xj(p2);
      float near2[2], mid[2], new2[2];
      wrap_to(x2, x1, near2);
      mid[0] = 0.5f * (x1[0] + near2[0]);
      mid[1] = 0.5f * (x1[1] + near2[1]);
      wrap_to(mid, x2, new2);
      S.put_f(&x1[0], mid[0]), S.put_f(&x1[1], mid[1]), S.put_f(&x2[0], new2[0]), S.put_f(&x2[1], new2[1]);
      bool flip = inside_triangle(x2, x1, xj(a1), xj(a2));
      // look ahead: junctions that later switches of this call still need keep their side
      auto later = [&](int64_t j) {
        for (size_t q = k; q < cols.size(); ++q)
          if (pp0[cols[q]] == j || pp1[cols[q]] == j) return true;
        return false;
      };
      const bool la1 = later(a1), lb1 = later(b1), la2 = later(a2), lb2 = later(b2);
      if (la2 && !lb2) flip = false;
      if (lb2 && !la2) flip = true;
      if (la1 && !lb1) flip = true;
      if (lb1 && !la1) flip = false;
      if (flip) {
        std::reverse(d1.begin(), d1.end()), std::reverse(d2.begin(), d2.end());
        std::reverse(e1.begin(), e1.end()), std::reverse(e2.begin(), e2.end());
        std::swap(a1, b1), std::swap(a2, b2);
      }
      set_grain(d1[1], grow2[0]);
      set_grain(d2[0], grow1[0]);
      S.put(0, e1[1], p2);
      S.put(0, e2[0], p1);
      ipp0.cols(a2, tmp);
      for (int64_t c : tmp)
        if (pp1[c] == p2) S.put(1, c, p1);
      ipp0.cols(b1, tmp);
      for (int64_t c : tmp)
        if (pp1[c] == p1) S.put(1, c, p2);
    }
    for (int64_t p : touched) {
      // KEEP (models.py:903, 1045-1047): the reference remembers a VIEW of the rewound position, so the displacement
      // feature of every touched junction comes out as 0 (x - x: NaN for a non-finite coordinate, as there).
      float *x = xj(p), *y = yj(p);
      S.put_f(&y[0], JOINT_SCALING * (x[0] - x[0]));
      S.put_f(&y[1], JOINT_SCALING * (x[1] - x[1]));
      if (A.ldx < 8) refuse("x_joint needs 8 feature columns");
      S.put_f(&x[6], y[0]), S.put_f(&x[7], y[1]);
    }
    return forced;
  }

  // -- models.py:628-717: shrink `grain` to two sides by switching all but two of its edges (those towards the neighbours
  // with the smallest predicted area change go first), then remove it.  false = the grain was skipped. --
  bool eliminate_grain(int64_t grain, std::vector<int64_t>& pending, std::vector<int64_t>& forced) {
    std::vector<int64_t> cc, corners, tmp;
    ipq1.cols(grain_id(grain), cc);
    for (int64_t c : cc) corners.push_back(pq0[c]);
    if (corners.empty()) return false;
    for (int64_t p : corners)
      if (!is_active(p)) return false;
    std::vector<int64_t> edge_cols, across;
    for (size_t i = 0; i < corners.size(); ++i)
      for (size_t j = i + 1; j < corners.size(); ++j) {   // itertools.combinations order
        const int64_t lo = std::min(corners[i], corners[j]), hi = std::max(corners[i], corners[j]);
        ipp0.cols(lo, tmp);
        size_t hits = 0;
        for (int64_t c : tmp)
          if (pp1[c] == hi) edge_cols.push_back(c), ++hits;
        if (hits == 0) continue;
        std::vector<int64_t> other_lo, other_hi;
        ipq0.cols(lo, tmp);
        for (int64_t c : tmp)
          if (pq1[c] != grain) other_lo.push_back(pq1[c]);
        ipq0.cols(hi, tmp);
        for (int64_t c : tmp)
          if (pq1[c] != grain) other_hi.push_back(pq1[c]);
        auto in = [&](int64_t g) { return std::find(other_hi.begin(), other_hi.end(), g) != other_hi.end(); };
        // (models.py:668-673 tests Nq1[0] first: a junction with ONE other grain passes when that grain is across the edge)
        if (other_lo.empty()) refuse("junction %lld of grain %lld has no other grain", (long long)lo, (long long)grain);
        if (in(other_lo[0])) across.push_back(other_lo[0]);
        else if (other_lo.size() >= 2 && in(other_lo[1])) across.push_back(other_lo[1]);
        else refuse("edge (%lld, %lld) of grain %lld has no grain on its other side", (long long)lo, (long long)hi, (long long)grain);
      }
    if (across.size() != corners.size())
      refuse("grain %lld: %zu junctions but %zu edges", (long long)grain, corners.size(), across.size());
    // (a junction pair joined by MORE than one column -- seen once, in a trajectory collapsing from 401 to 32 grains in a step --
    // leaves edge_cols longer than across; the reference indexes the concatenated hits with the order of `across` all the same
    // (models.py: edge_cols[order[:-2]]): kept, as the scan oracle keeps it.  Round 6 refused here; the two implementations then
    // refused one step apart.)
    {
      std::vector<int64_t> s(across);
      std::sort(s.begin(), s.end());
      if (std::unique(s.begin(), s.end()) != s.end()) return false;
    }
    // np.argsort(y_grain_area[across], kind="stable"): ascending, NaN last
    std::vector<size_t> order(across.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    auto area = [&](size_t i) { return A.y_grain_area[grain_id(across[i]) * A.ldyg]; };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
      const float u = area(a), v = area(b);
      return (u < v) || (std::isnan(v) && !std::isnan(u));
    });
    std::vector<int64_t> cols;
    for (size_t i = 0; i + 2 < order.size(); ++i) cols.push_back(edge_cols[order[i]]);
    forced = switch_edges(cols, true, grain);
    remove_two_sided_grain(grain);
    for (int64_t g : forced) remove_two_sided_grain(g);
    for (int64_t c : cols) {
      auto it = std::find(pending.begin(), pending.end(), c);
      if (it != pending.end()) pending.erase(it);
    }
    remove_all_two_sided();
    return true;
  }

  void run() {
    const float thr = (float)A.threshold;   // (numpy compares the fp32 probabilities with the Python float in fp32)
    // the caller's column c is the session's c-th live stable column (same relative order)
    std::vector<int64_t> pending, extra, forced;
    std::unordered_map<int64_t, float> prob_of;   // probabilities of the pending columns, by stable column
    const int64_t n_ext = S.n_live[0];
    for (int64_t c = 0; c < n_ext; ++c) {
      if (!(A.edge_prob[c] > thr)) continue;
      const int64_t k = S.kth_live_pp(c);
      if (pp0[(size_t)k] < pp1[(size_t)k]) pending.push_back(k), prob_of[k] = A.edge_prob[c];
    }
    // the output lists' room is checked BEFORE anything is rewritten (the switching list only shrinks from here on; every
    // grain and the dead-column marker can be reported at most once)
    if ((int64_t)pending.size() > A.switching_cap) refuse("switching list: room for %lld pairs needed", (long long)pending.size());
    if (A.extra_cap < A.n_grain + 1) refuse("event list: room for n_grain + 1 = %lld grains needed", (long long)(A.n_grain + 1));
    for (int64_t k = 0; k < A.n_grain_event; ++k) {
      const int64_t g = grain_id(A.grain_event[k]);
      if (A.active_grain != nullptr && !A.active_grain[g]) continue;
      forced.clear();
      if (eliminate_grain(g, pending, forced)) extra.insert(extra.end(), forced.begin(), forced.end());
    }
    // neighbour switching, most probable edge first (ties: lower column first): argsort(-prob, stable), NaN last
    std::stable_sort(pending.begin(), pending.end(), [&](int64_t a, int64_t b) {
      const float u = -prob_of[a], v = -prob_of[b];
      return (u < v) || (std::isnan(v) && !std::isnan(u));
    });
    pending.erase(std::remove_if(pending.begin(), pending.end(), [&](int64_t c) { return pp0[(size_t)c] == DEAD; }), pending.end());
    switch_edges(pending, false, DEAD);
    A.n_switching = (int64_t)pending.size();
    for (size_t i = 0; i < pending.size(); ++i)
      A.switching[2 * i] = pp0[(size_t)pending[i]], A.switching[2 * i + 1] = pp1[(size_t)pending[i]];
    const std::vector<int64_t> last = remove_all_two_sided();
    extra.insert(extra.end(), last.begin(), last.end());
    if ((int64_t)extra.size() > A.extra_cap) refuse("event list: room for %lld more grains needed", (long long)extra.size());
    A.n_extra = (int64_t)extra.size();
    std::copy(extra.begin(), extra.end(), A.events_extra);
  }
};

int check_apply_args(const ggnn_topology_args& A) {
  if (!A.x_joint || !A.y_joint || !A.y_grain_area || !A.edge_prob || !A.mask_grain || !A.mask_joint ||
      (A.n_grain_event > 0 && !A.grain_event) || !A.switching || !A.events_extra)
    return GGNN_EINVAL;
  if (A.n_joint <= 0 || A.n_grain <= 0 || A.ldx < 8 || A.ldyg < 1 || A.n_grain_event < 0 || A.switching_cap < 0 || A.extra_cap < 0)
    return GGNN_EINVAL;
  return GGNN_OK;
}

// Validates and loads the lists; the message of a refusal goes to `error` (192 bytes) when given.
int open_session(const int64_t* pp0, const int64_t* pp1, int64_t n_pp, const int64_t* pq0, const int64_t* pq1, int64_t n_pq,
                 int64_t n_joint, int64_t n_grain, Session** out, char* error) {
  if (!out || n_pp < 0 || n_pq < 0 || n_joint <= 0 || n_grain <= 0 || (n_pp > 0 && (!pp0 || !pp1)) || (n_pq > 0 && (!pq0 || !pq1)))
    return GGNN_EINVAL;
  auto bad = [&](const char* what, int64_t v, int64_t n) {
    if (error) snprintf(error, 192, "%s index %lld outside [0, %lld)", what, (long long)v, (long long)n);
    return GGNN_ETOPOLOGY;
  };
  for (int64_t c = 0; c < n_pp; ++c) {
    if (pp0[c] < 0 || pp0[c] >= n_joint) return bad("junction", pp0[c], n_joint);
    if (pp1[c] < 0 || pp1[c] >= n_joint) return bad("junction", pp1[c], n_joint);
  }
  for (int64_t c = 0; c < n_pq; ++c) {
    if (pq0[c] < 0 || pq0[c] >= n_joint) return bad("junction", pq0[c], n_joint);
    if (pq1[c] < 0 || pq1[c] >= n_grain) return bad("grain", pq1[c], n_grain);
  }
  try {
    Session* s = new Session();
    s->n_joint = n_joint, s->n_grain = n_grain;
    s->load(pp0, pp1, n_pp, pq0, pq1, n_pq);
    *out = s;
  } catch (const std::exception& e) {
    if (error) snprintf(error, 192, "%s", e.what());
    return GGNN_ETOPOLOGY;
  }
  return GGNN_OK;
}

int apply_session(Session& S, ggnn_topology_args& A) {
  A.error[0] = 0;
  A.n_switching = A.n_extra = 0;
  if (const int rc = check_apply_args(A)) return rc;
  if (A.n_joint != S.n_joint || A.n_grain != S.n_grain) return GGNN_EINVAL;
  S.commit();
  try {
    Topology T(A, S);
    T.run();
    S.commit();
    S.rebase_if_sparse();   // (models.py:845-858 drops the dead columns, order kept -- here the caller's numbering skips them)
  } catch (const Refused&) {
    S.rollback();
    return GGNN_ETOPOLOGY;
  } catch (const std::exception& e) {
    S.rollback();
    snprintf(A.error, sizeof(A.error), "%s", e.what());
    return GGNN_ETOPOLOGY;
  }
  A.n_pp = S.n_live[0], A.n_pq = S.n_live[1];
  return GGNN_OK;
}

}  // namespace

extern "C" int ggnn_topology_open(const int64_t* pp, int64_t n_pp, int64_t pp_ld, const int64_t* pq, int64_t n_pq,
                                  int64_t pq_ld, int64_t n_joint, int64_t n_grain, ggnn_topology_session** session,
                                  char* error) {
  if (error) error[0] = 0;
  if (pp_ld < n_pp || pq_ld < n_pq) return GGNN_EINVAL;
  return open_session(pp, pp ? pp + pp_ld : nullptr, n_pp, pq, pq ? pq + pq_ld : nullptr, n_pq, n_joint, n_grain, session, error);
}

extern "C" void ggnn_topology_close(ggnn_topology_session* session) { delete session; }

extern "C" int ggnn_topology_apply(ggnn_topology_session* session, ggnn_topology_args* args) {
  if (!session || !args) return GGNN_EINVAL;
  return apply_session(*session, *args);
}

extern "C" int ggnn_topology_counts(const ggnn_topology_session* session, int64_t* n_pp, int64_t* n_pq) {
  if (!session || !n_pp || !n_pq) return GGNN_EINVAL;
  *n_pp = session->n_live[0], *n_pq = session->n_live[1];
  return GGNN_OK;
}

extern "C" int ggnn_topology_export(const ggnn_topology_session* session, int64_t* pp, int64_t pp_ld, int64_t* pq,
                                    int64_t pq_ld, int64_t* qp, int64_t qp_ld) {
  if (!session) return GGNN_EINVAL;
  const Session& S = *session;
  const int64_t n_pp = S.n_live[0], n_pq = S.n_live[1];
  if ((pp && pp_ld < n_pp) || (pq && pq_ld < n_pq) || (qp && qp_ld < n_pq)) return GGNN_EINVAL;
  // one pass over the stable columns, the live ones written out in order
  if (pp) {
    const int64_t *a0 = S.arr[0].data(), *a1 = S.arr[1].data();
    int64_t w = 0;
    for (size_t k = 0, n = S.arr[0].size(); k < n; ++k)
      if (a0[k] != -1) pp[w] = a0[k], pp[pp_ld + w] = a1[k], ++w;
  }
  if (pq || qp) {
    const int64_t *a0 = S.arr[2].data(), *a1 = S.arr[3].data();
    int64_t w = 0;
    for (size_t k = 0, n = S.arr[2].size(); k < n; ++k) {
      if (a0[k] == -1) continue;
      if (pq) pq[w] = a0[k], pq[pq_ld + w] = a1[k];
      if (qp) qp[w] = a1[k], qp[qp_ld + w] = a0[k];
      ++w;
    }
  }
  return GGNN_OK;
}

// The stateless form: a session for one call on the caller's arrays.
extern "C" int ggnn_topology_update(ggnn_topology_args* args) {
  if (!args) return GGNN_EINVAL;
  ggnn_topology_args& A = *args;
  A.error[0] = 0;
  A.n_switching = A.n_extra = 0;
  if (!A.pp || !A.pq) return GGNN_EINVAL;
  if (const int rc = check_apply_args(A)) return rc;
  if (A.n_pp < 0 || A.n_pq < 0 || A.pp_cap < A.n_pp || A.pq_cap < A.n_pq) return GGNN_EINVAL;
  Session* s = nullptr;
  int rc = open_session(A.pp, A.pp + A.pp_cap, A.n_pp, A.pq, A.pq + A.pq_cap, A.n_pq, A.n_joint, A.n_grain, &s, A.error);
  if (rc != GGNN_OK) return rc;
  rc = apply_session(*s, A);
  if (rc == GGNN_OK) {
    if (A.n_pp > A.pp_cap || A.n_pq > A.pq_cap) {
      snprintf(A.error, sizeof(A.error), "the junction edge list needs room for %lld columns (pp_cap)", (long long)A.n_pp);
      rc = GGNN_ETOPOLOGY;
    } else {
      rc = ggnn_topology_export(s, A.pp, A.pp_cap, A.pq, A.pq_cap, nullptr, 0);
    }
  }
  delete s;
  return rc;
}
