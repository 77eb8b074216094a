// Host-only arithmetic of the buffer-placement allocator (mdc_placement.hip), in a header both compilers take: how a set of probe times is
// split into a fast and a slow cluster, how groups of pieces get their memory class from two such splits, and which pieces make up
// which range.  tests/native/placement_classes_cpu.cpp (tests/test_placement_cpu.py) pins it on the CPU with synthetic times.
#pragma once
#include <algorithm>
#include <cstddef>
#include <limits>
#include <vector>

namespace mdc {

// Two clusters of times: a pair of memory pieces in one class runs 5-9 % slower than a pair across classes, the noise inside a cluster is
// ~1 %.  The split of the sorted times that maximises the between-cluster variance (Otsu), neither side smaller than a tenth of the set
// (stray fast or slow measurements must not become a "class").  -> the threshold between the clusters, or +inf when their means are
// less than 3 % apart (one class as far as can be seen); *rel = how far apart the means are, relative to the fast one.
inline float placement_cut(std::vector<float> v, float* rel) {
  *rel = 0.f;
  const size_t n = v.size();
  if (n < 2) return std::numeric_limits<float>::infinity();
  std::sort(v.begin(), v.end());
  std::vector<double> pre(n + 1, 0.0);
  for (size_t i = 0; i < n; i++) pre[i + 1] = pre[i] + v[i];
  const size_t minsz = std::max<size_t>(1, n / 10);
  double best = -1;
  size_t at = 0;
  for (size_t i = minsz; i + minsz <= n; i++) {
    const double m1 = pre[i] / i, m2 = (pre[n] - pre[i]) / (n - i), sc = (double)i * (n - i) * (m2 - m1) * (m2 - m1);
    if (sc > best) best = sc, at = i;
  }
  if (!at) return std::numeric_limits<float>::infinity();
  const double m1 = pre[at] / at, m2 = (pre[n] - pre[at]) / (n - at);
  *rel = m1 > 0 ? (float)((m2 - m1) / m1) : 0.f;
  return *rel > 0.03f ? 0.5f * (v[at - 1] + v[at]) : std::numeric_limits<float>::infinity();
}

// Classes of the groups from their times against two references.  t0[g] = stream time of group g against reference group ref0 (t0[ref0]
// unused); the slow cluster is ref0's own class (0).  Of the fast ones, the first is the second reference ref1 (given, or chosen here when
// *ref1 < 0); t1[g] = time against it (< 0: not measured yet -- then `need` lists the groups still to be timed and the classes of the fast
// groups are provisional): slow with ref1 = class 1, fast with both = class 2.  -> class per group.
inline std::vector<int> placement_classes(const std::vector<float>& t0, size_t ref0, const std::vector<float>& t1, long* ref1, float rel[2],
                                          std::vector<size_t>* need) {
  const size_t n = t0.size();
  std::vector<int> cls(n, 0);
  rel[0] = rel[1] = 0.f;
  if (need) need->clear();
  if (n < 2) return cls;
  std::vector<float> v0;
  for (size_t g = 0; g < n; g++)
    if (g != ref0) v0.push_back(t0[g]);
  const float cut0 = placement_cut(v0, &rel[0]);
  std::vector<size_t> fast0;
  for (size_t g = 0; g < n; g++)
    if (g != ref0 && cut0 < std::numeric_limits<float>::infinity() && t0[g] < cut0) fast0.push_back(g);
  if (fast0.empty()) return cls;
  if (*ref1 < 0) *ref1 = (long)fast0[0];
  std::vector<float> v1;
  for (size_t g : fast0) {
    if ((long)g == *ref1) continue;
    if (g >= t1.size() || t1[g] < 0) {
      if (need) need->push_back(g);
    } else {
      v1.push_back(t1[g]);
    }
  }
  const float cut1 = placement_cut(v1, &rel[1]);
  for (size_t g : fast0) {
    const bool timed = g < t1.size() && t1[g] >= 0;
    cls[g] = ((long)g == *ref1 || !timed || !(cut1 < std::numeric_limits<float>::infinity() && t1[g] < cut1)) ? 1 : 2;
  }
  return cls;
}

// Which pieces make up which range: piece by piece round the classes, all ranges in step, range k starting at class k mod 3 (a frame and
// its result lie at about the same relative position of their ranges); a class that has run out passes its turn to the next.
// by_class[c] = the pieces of class c in creation order; want[k] = pieces of range k.  -> piece ids per range.
inline std::vector<std::vector<size_t>> placement_compose(const std::vector<size_t> by_class[3], const std::vector<size_t>& want) {
  std::vector<std::vector<size_t>> out(want.size());
  size_t at[3] = {0, 0, 0};
  std::vector<int> turn(want.size());
  size_t longest = 0;
  for (size_t i = 0; i < want.size(); i++) turn[i] = (int)(i % 3), longest = std::max(longest, want[i]);
  for (size_t k = 0; k < longest; k++)
    for (size_t i = 0; i < want.size(); i++) {
      if (k >= want[i]) continue;
      for (int q = 0; q < 3; q++) {
        const int c = (turn[i] + q) % 3;
        if (at[c] < by_class[c].size()) {
          out[i].push_back(by_class[c][at[c]++]);
          turn[i] = (c + 1) % 3;
          break;
        }
      }
    }
  return out;
}

}  // namespace mdc
