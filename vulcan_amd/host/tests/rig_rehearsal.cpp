// rig_rehearsal.cpp — the rig's in-launch exchange (vulcan_amd/csrc/vk_rig_protocol.h) on the CPU:
// `world` host threads stand in for the ranks' publishing workgroups, std::atomic words in host
// memory for the peer-mapped areas. Each "rank" runs `steps` Gauss-Newton steps per Track and
// `tracks` Tracks: publish 27 values derived from (rank, track, step, word), gather, and check that
// the total is the rank-ordered float sum of exactly THIS step's values on every rank — i.e. that
// tags, slot indexing and the two-parity buffering never hand a reader a value of another step,
// whatever the threads' relative speed (random pauses). Exit code 0 = all checks passed.
//
//   rig_rehearsal [world=2] [tracks=50] [steps=200] [seed=1]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../../csrc/vk_rig_protocol.h"

static float value_of(int rank, int track, int step, int word)
{
  // distinct per (rank, track, step, word); sums are order-sensitive in float32
  return 1.0f + 0.37f * rank + 1e-3f * track + 1e-5f * step + 0.011f * word + ((rank * 7 + step * 3 + word) % 5) * 1e-7f;
}

int main(int argc, char** argv)
{
  const int world = argc > 1 ? atoi(argv[1]) : 2;
  const int tracks = argc > 2 ? atoi(argv[2]) : 50;
  const int steps = argc > 3 ? atoi(argv[3]) : 200;
  const unsigned seed = argc > 4 ? (unsigned)atoi(argv[4]) : 1u;
  if (world < 1 || world > VK_RIG_MAX_RANKS) return 2;

  std::vector<std::vector<std::atomic<unsigned long long>>> areas(world);
  for (auto& a : areas) { a = std::vector<std::atomic<unsigned long long>>(rig_area_words()); for (auto& w : a) w.store(0); }
  std::vector<unsigned long long*> raw(world);
  for (int r = 0; r < world; ++r) raw[r] = reinterpret_cast<unsigned long long*>(areas[r].data());

  std::atomic<int> failures(0);
  auto store = [](unsigned long long* at, unsigned long long w) {
    reinterpret_cast<std::atomic<unsigned long long>*>(at)->store(w, std::memory_order_relaxed);
  };
  auto load = [](const unsigned long long* at) {
    return reinterpret_cast<const std::atomic<unsigned long long>*>(at)->load(std::memory_order_relaxed);
  };

  auto rank_main = [&](int rank) {
    std::mt19937 rng(seed * 977u + (unsigned)rank);
    for (int track = 0; track < tracks; ++track)
    {
      const uint32_t sequence = (uint32_t)(track + 1);       // the same on every rank, never 0
      for (int step = 0; step < steps; ++step)
      {
        if (rng() % 7 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 50));
        for (int word = 0; word < VK_RIG_VALUES; ++word)
          rig_publish(raw.data(), rank, world, sequence, step, word, value_of(rank, track, step, word), store);
        if (rng() % 11 == 0) std::this_thread::yield();
        for (int word = 0; word < VK_RIG_VALUES; ++word)
        {
          float total = 0.0f;
          const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(20);
          const bool ok = rig_gather(raw[rank], world, sequence, step, word, total, load,
              [&] { return std::chrono::steady_clock::now() > deadline; });
          float want = value_of(0, track, step, word);
          for (int s = 1; s < world; ++s) want = want + value_of(s, track, step, word);
          if (!ok || total != want)
          {
            if (failures.fetch_add(1) < 5)
              std::fprintf(stderr, "rank %d track %d step %d word %d: got %.9g want %.9g%s\n", rank, track, step, word,
                  (double)total, (double)want, ok ? "" : " (timed out)");
          }
        }
      }
    }
  };

  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) threads.emplace_back(rank_main, r);
  for (auto& t : threads) t.join();
  std::printf("rig_rehearsal: world %d, %d tracks x %d steps x %d words, %d failure(s)\n", world, tracks, steps,
      VK_RIG_VALUES, failures.load());
  return failures.load() ? 1 : 0;
}
