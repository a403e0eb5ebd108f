// Prints what accumulation_amd/csrc/msm_select.h decides at every threshold edge (plain C++: no HIP, no library) --
// tests/test_pipeline_select_cpu.py holds the expected table.  One line per (key, pairs, form):
//   <key> <pairs> <plain|grouped|grouped_irregular|skewed> <pipeline> twin=<0|1> plain_window=<c> range=<pairs>
#include <cstdio>
#include <vector>

#include "../../accumulation_amd/csrc/msm_select.h"

using namespace amsm::msel;

int main() {
  struct K {
    const char* name;
    KeyDesc d;
  };
  const std::vector<K> keys = {
      {"direct_2p15", {P2(15), true, false, true, 13}},    // precomputed, carries the direct-sum table
      {"table_2p16", {P2(16), true, false, false, 16}},    // precomputed, 16-bit windows
      {"table_2p19", {P2(19), true, false, false, 16}},
      {"bpl_2p20", {P2(20), true, true, false, 20}},       // the 20-bit table
      {"bpl_2p22", {P2(22), true, true, false, 20}},
      {"plain_2p22", {P2(22), false, false, false, 0}},
  };
  std::vector<size_t> ns;
  for (int lg : {15, 16, 17, 18, 19, 20}) {
    ns.push_back(P2(lg) - 1);
    ns.push_back(P2(lg));
    ns.push_back(P2(lg) + 1);
  }
  ns.push_back(1);
  ns.push_back(P2(21));
  ns.push_back(P2(22) - 1);
  ns.push_back(P2(22));
  Switches sw;
  for (const K& k : keys)
    for (size_t n : ns) {
      if (n > k.d.n) continue;
      struct F {
        const char* name;
        bool grouped, regular, skewed;
      } forms[] = {{"plain", false, false, false}, {"grouped", true, true, false}, {"grouped_irregular", true, false, false},
                   {"skewed", false, false, true}};
      for (const F& f : forms) {
        const Choice c = choose(k.d, n, f.grouped, f.regular, f.skewed, sw);
        printf("%s %zu %s %s twin=%d plain_window=%d range=%zu probe=%d\n", k.name, n, f.name, pipeline_name(c.pipeline), c.over_twin ? 1 : 0,
               c.plain_window, c.range, wants_skew_probe(k.d, n, sw) ? 1 : 0);
      }
    }
  // the switches
  Switches off = sw;
  off.bpl = false;
  printf("switch bpl=0 bpl_2p20 %s\n", pipeline_name(choose(keys[3].d, P2(20), false, false, false, off).pipeline));
  off = sw;
  off.bpl_plain = false;
  printf("switch bpl_plain=0 plain %s range=%zu\n", pipeline_name(choose(keys[5].d, P2(20), false, false, false, off).pipeline), range_of(keys[5].d, P2(22), off));
  off = sw;
  off.bps = 0;
  printf("switch bps=0 table_2p16 %s\n", pipeline_name(choose(keys[1].d, P2(16), false, false, false, off).pipeline));
  off.bps = 1;
  printf("switch bps=1 table_2p16 plain %s grouped %s\n", pipeline_name(choose(keys[1].d, P2(16), false, false, false, off).pipeline),
         pipeline_name(choose(keys[1].d, P2(16), true, true, false, off).pipeline));
  off = sw;
  off.direct = false;
  printf("switch direct=0 direct_2p15 %s\n", pipeline_name(choose(keys[0].d, P2(12), false, false, false, off).pipeline));
  off = sw;
  off.window_override = true;
  printf("switch window_override bpl_2p20 %s twin=%d plain %s\n", pipeline_name(choose(keys[3].d, P2(20), false, false, false, off).pipeline),
         choose(keys[3].d, P2(20), false, false, false, off).over_twin ? 1 : 0, pipeline_name(choose(keys[5].d, P2(20), false, false, false, off).pipeline));
  off = sw;
  off.split_log2 = 0;
  K big{"table_2p23", {P2(23), true, false, false, 17}};
  printf("split table_2p23 default range=%zu off range=%zu below range=%zu\n", range_of(big.d, P2(23), sw), range_of(big.d, P2(23), off),
         range_of(big.d, P2(22) - 1, sw));
  // table 1: the window a key is built for
  for (int lg : {1, 8, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22})
    printf("key_window 2p%d precomputed=%d precomputed_bpl_off=%d plain=%d\n", lg, key_window(P2(lg), true, true), key_window(P2(lg), true, false),
           key_window(P2(lg), false, true));
  return 0;
}
