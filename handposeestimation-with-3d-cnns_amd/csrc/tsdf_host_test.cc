// Sanitizer build of the library's host-only code (tsdf_host.inc) — test infrastructure, never shipped:
//   make -C handposeestimation-with-3d-cnns_amd/csrc host-asan   ->  build/libtsdf_host_asan.so  (-fsanitize=address,undefined)
//   make -C handposeestimation-with-3d-cnns_amd/csrc host-tsan   ->  build/libtsdf_host_tsan.so  (-fsanitize=thread)
// The same source text the product compiles, plus a few hooks so that tests/test_tiers_cpu.py can drive the slot table
// and the argument checks, which have no C entry of their own in the product.
#include "tsdf_host.inc"

#include <atomic>
#include <vector>

namespace {
tsdf_host::SlotTable<64> g_table;   // (a small table: "full" is reachable in a test)
}

extern "C" {

// tsdf_host::check_run_args with the product's defaults for what the hook does not take
int tsdf_test_check_run_args(const void *d_depth, int64_t depth_len, const void *d_offsets, const void *d_headers, int n, int R,
                             const tsdf_cam *cam, int layout, const void *out_tsdf, int aabb_only, const tsdf_labels *labels) {
  const int supported = R >= 4 && R <= 128 && (R % 4) == 0;
  return tsdf_host::check_run_args(d_depth, depth_len, d_offsets, d_headers, n, R, cam, layout, out_tsdf, aabb_only, labels,
                                   supported);
}

int tsdf_test_slot_acquire(uintptr_t stream, int per_thread) {
  return g_table.acquire(reinterpret_cast<const void *>(stream), per_thread != 0);
}
void tsdf_test_slot_release(uintptr_t stream, int per_thread) {
  g_table.release(reinterpret_cast<const void *>(stream), per_thread != 0);
}
uint32_t tsdf_test_slot_next_epoch(int slot) { return g_table.next_epoch(slot); }

// `threads` threads hammer the table: each owns `per` private streams (acquire, check the slot is stable and not shared,
// draw epochs, release, re-acquire) and all of them share one stream value through the per-thread key.  Returns the
// number of violated invariants (0 = fine); under -fsanitize=thread any unsynchronised access aborts the process.
int tsdf_test_slot_hammer(int threads, int per, int rounds) {
  std::atomic<int> bad{0};
  std::vector<std::atomic<uintptr_t>> owner(64);
  for (auto &o : owner) o.store(0);
  auto work = [&](int t) {
    for (int r = 0; r < rounds; ++r) {
      for (int k = 0; k < per; ++k) {
        const uintptr_t s = 0x1000 + (uintptr_t)(t * per + k) * 16;
        const int i = g_table.acquire(reinterpret_cast<const void *>(s), false);
        if (i < 0) continue;                                   // table full: allowed
        uintptr_t expect = 0;
        if (!owner[i].compare_exchange_strong(expect, s) && expect != s) ++bad;   // two live streams on one slot
        if (g_table.acquire(reinterpret_cast<const void *>(s), false) != i) ++bad;  // the pair's slot is stable
        const uint32_t e1 = g_table.next_epoch(i), e2 = g_table.next_epoch(i);
        if (e1 == 0 || e2 == 0 || e1 == e2) ++bad;             // consecutive launches never share an epoch
        owner[i].store(0);
        g_table.release(reinterpret_cast<const void *>(s), false);
      }
      // the per-thread key: one stream value, a different slot per thread while both hold it
      const int mine = g_table.acquire(reinterpret_cast<const void *>(0x2), true);
      if (mine >= 0) {
        if (g_table.acquire(reinterpret_cast<const void *>(0x2), true) != mine) ++bad;
        g_table.release(reinterpret_cast<const void *>(0x2), true);
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
  for (auto &th : pool) th.join();
  return bad.load();
}

}  // extern "C"
