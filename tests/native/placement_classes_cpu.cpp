// The allocator's host arithmetic (mono_dataset_code_amd/csrc/placement_classes.h) on synthetic probe times: g++, no GPU.
// Prints one line per case "name ok" / "name FAIL ..." and exits non-zero on any failure (tests/test_placement_cpu.py).
#include <cmath>
#include <cstdio>
#include <random>
#include <set>

#include "placement_classes.h"

static int failures = 0;
#define EXPECT(name, cond)                                     \
  do {                                                         \
    if (cond) std::printf("%s ok\n", name);                    \
    else std::printf("%s FAIL (line %d)\n", name, __LINE__), failures++; \
  } while (0)

int main() {
  std::mt19937 rng(12345);
  std::uniform_real_distribution<float> jit(-0.01f, 0.01f);
  float rel = 0;
  {  // two clusters 7 % apart, 1 % noise
    std::vector<float> v;
    for (int i = 0; i < 20; i++) v.push_back(0.280f * (1 + jit(rng)));
    for (int i = 0; i < 10; i++) v.push_back(0.300f * (1 + jit(rng)));
    const float cut = mdc::placement_cut(v, &rel);
    int slow = 0;
    for (float x : v) slow += x > cut;
    EXPECT("bimodal_7pct", std::isfinite(cut) && slow == 10 && rel > 0.05f && rel < 0.09f);
  }
  {  // one cluster
    std::vector<float> v;
    for (int i = 0; i < 40; i++) v.push_back(0.290f * (1 + jit(rng)));
    EXPECT("unimodal", !std::isfinite(mdc::placement_cut(v, &rel)) && rel < 0.03f);
  }
  {  // 264 groups, a third slow, plus stray measurements 10 % fast and 25 % slow: strays must not become a class
    std::vector<float> v;
    for (int i = 0; i < 176; i++) v.push_back(0.275f * (1 + jit(rng)));
    for (int i = 0; i < 84; i++) v.push_back(0.292f * (1 + jit(rng)));
    v.push_back(0.248f), v.push_back(0.250f), v.push_back(0.36f), v.push_back(0.37f);
    const float cut = mdc::placement_cut(v, &rel);
    int slow = 0;
    for (float x : v) slow += x > cut;
    EXPECT("strays_are_no_class", std::isfinite(cut) && slow >= 84 && slow <= 88 && cut > 0.279f && cut < 0.289f);
  }
  {  // the smallest sets
    EXPECT("two_values_apart", std::isfinite(mdc::placement_cut({0.28f, 0.30f}, &rel)));
    EXPECT("two_values_close", !std::isfinite(mdc::placement_cut({0.28f, 0.283f}, &rel)));
    EXPECT("one_value", !std::isfinite(mdc::placement_cut({0.28f}, &rel)));
    EXPECT("no_value", !std::isfinite(mdc::placement_cut({}, &rel)));
  }
  {  // three classes A A B C A B B C C A: reference 0 is of class A; against it A is slow; of the fast ones the first (group 2, class B) is the second reference
    const int truth[10] = {0, 0, 1, 2, 0, 1, 1, 2, 2, 0};
    std::vector<float> t0(10), t1;
    for (int g = 0; g < 10; g++) t0[g] = (truth[g] == 0 ? 0.300f : 0.280f) * (1 + 0.3f * jit(rng));
    long ref1 = -1;
    float r2[2];
    std::vector<size_t> need;
    std::vector<int> cls = mdc::placement_classes(t0, 0, t1, &ref1, r2, &need);
    EXPECT("second_reference_is_first_fast_group", ref1 == 2 && need.size() == 5);  // groups 3 5 6 7 8 still to be timed against it
    t1.assign(10, -1.f);
    for (size_t g : need) t1[g] = (truth[g] == 1 ? 0.300f : 0.280f) * (1 + 0.3f * jit(rng));
    cls = mdc::placement_classes(t0, 0, t1, &ref1, r2, &need);
    bool same = need.empty();
    for (int g = 0; g < 10; g++) same = same && cls[g] == truth[g];
    EXPECT("three_classes", same && r2[0] > 0.05f && r2[1] > 0.05f);
  }
  {  // everything in the reference's class
    std::vector<float> t0(8, 0.3f), t1;
    t0[0] = 0;
    long ref1 = -1;
    float r2[2];
    std::vector<size_t> need;
    const std::vector<int> cls = mdc::placement_classes(t0, 0, t1, &ref1, r2, &need);
    bool all0 = ref1 == -1 && need.empty();
    for (int x : cls) all0 = all0 && x == 0;
    EXPECT("one_class", all0);
  }
  {  // composition: 11 + 10 pieces from classes of 7 / 3 / 12: every piece once, range 0 starts in class 0, range 1 in class 1, a class that ran out passes its turn
    std::vector<size_t> by[3];
    for (size_t k = 0; k < 7; k++) by[0].push_back(k);
    for (size_t k = 0; k < 3; k++) by[1].push_back(100 + k);
    for (size_t k = 0; k < 12; k++) by[2].push_back(200 + k);
    const auto ids = mdc::placement_compose(by, {11, 10});
    std::set<size_t> seen;
    for (const auto& r : ids) seen.insert(r.begin(), r.end());
    auto cls_of = [](size_t id) { return id >= 200 ? 2 : id >= 100 ? 1 : 0; };
    int share0[3] = {0, 0, 0}, share1[3] = {0, 0, 0};
    for (size_t id : ids[0]) share0[cls_of(id)]++;
    for (size_t id : ids[1]) share1[cls_of(id)]++;
    EXPECT("compose_counts", ids.size() == 2 && ids[0].size() == 11 && ids[1].size() == 10 && seen.size() == 21);
    EXPECT("compose_starts", cls_of(ids[0][0]) == 0 && cls_of(ids[1][0]) == 1 && cls_of(ids[0][1]) == 1 && cls_of(ids[1][1]) == 2);
    EXPECT("compose_shares", share0[1] + share1[1] == 3 && share0[0] >= 3 && share1[0] >= 3 && share0[2] >= 4 && share1[2] >= 4);
  }
  return failures ? 1 : 0;
}
