// rig_rehearsal.cpp — the rig's in-launch exchange (vulcan_amd/csrc/vk_rig_protocol.h) on the CPU:
// `world` host threads stand in for the ranks' publishing workgroups, std::atomic words in host
// memory for the peer-mapped areas. Each "rank" runs `steps` Gauss-Newton steps per Track and
// `tracks` Tracks: publish 27 values derived from (rank, track, step, word), gather, and check that
// the total is the rank-ordered float sum of exactly THIS step's values on every rank — i.e. that
// tags, slot indexing and the two-parity buffering never hand a reader a value of another step,
// whatever the threads' relative speed (random pauses). Exit code 0 = all checks passed.
//
// Two more things the ranks' hosts do are rehearsed with it: the sequence number moves on by rig_next_sequence
// after EVERY Track (start_sequence puts the wrap 2^22 - 2 -> 1 inside the run), and every abort_every-th Track is
// ABORTED — one rank stops half way through a step and publishes no more, the others give up waiting — after which
// all ranks meet (the hosts' agreement, vulcan_amd/comm.py Communicator.agree), move on to the next number and must
// find the following Tracks' sums intact: words of the aborted attempt stay behind under tags nobody asks for again.
//
//   rig_rehearsal [world=2] [tracks=50] [steps=200] [seed=1] [start_sequence=1] [abort_every=0]
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
  const uint32_t start_sequence = argc > 5 ? (uint32_t)strtoul(argv[5], nullptr, 10) : 1u;
  const int abort_every = argc > 6 ? atoi(argv[6]) : 0;
  if (world < 1 || world > VK_RIG_MAX_RANKS || start_sequence < 1 || start_sequence > VK_RIG_LAST_SEQUENCE) return 2;

  std::vector<std::vector<std::atomic<unsigned long long>>> areas(world);
  for (auto& a : areas) { a = std::vector<std::atomic<unsigned long long>>(rig_area_words()); for (auto& w : a) w.store(0); }
  std::vector<unsigned long long*> raw(world);
  for (int r = 0; r < world; ++r) raw[r] = reinterpret_cast<unsigned long long*>(areas[r].data());

  std::atomic<int> failures(0), aborted_tracks(0), met(0);
  // the hosts' meeting after an aborted Track (a collective over their own channel): a counting barrier
  auto meet = [&](int round) {
    met.fetch_add(1);
    while (met.load() < world * (round + 1)) std::this_thread::yield();
  };
  auto store = [](unsigned long long* at, unsigned long long w) {
    reinterpret_cast<std::atomic<unsigned long long>*>(at)->store(w, std::memory_order_relaxed);
  };
  auto load = [](const unsigned long long* at) {
    return reinterpret_cast<const std::atomic<unsigned long long>*>(at)->load(std::memory_order_relaxed);
  };

  auto rank_main = [&](int rank) {
    std::mt19937 rng(seed * 977u + (unsigned)rank);
    uint32_t sequence = start_sequence;                      // the same on every rank, never 0
    int meetings = 0;
    for (int track = 0; track < tracks; ++track, sequence = rig_next_sequence(sequence))
    {
      // an aborted Track: `quitter` publishes half the words of step `quit_step` and leaves; the others wait 30 ms
      const bool doomed = abort_every > 0 && world > 1 && track % abort_every == abort_every - 1;
      const int quitter = track % world, quit_step = steps / 2;
      bool left = false;
      for (int step = 0; step < steps && !left; ++step)
      {
        if (rng() % 7 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 50));
        const bool quitting = doomed && rank == quitter && step == quit_step;
        for (int word = 0; word < (quitting ? VK_RIG_VALUES / 2 : VK_RIG_VALUES); ++word)
          rig_publish(raw.data(), rank, world, sequence, step, word, value_of(rank, track, step, word), store);
        if (quitting) { left = true; break; }
        if (rng() % 11 == 0) std::this_thread::yield();
        for (int word = 0; word < VK_RIG_VALUES && !left; ++word)
        {
          float total = 0.0f;
          const bool may_starve = doomed && step >= quit_step;
          const auto deadline = std::chrono::steady_clock::now() + (may_starve ? std::chrono::milliseconds(30) : std::chrono::milliseconds(20000));
          const bool ok = rig_gather(raw[rank], world, sequence, step, word, total, load,
              [&] { return std::chrono::steady_clock::now() > deadline; });
          if (!ok && may_starve) { left = true; break; }     // VK_TRACK_ABORTED: no pose, the host moves on
          float want = value_of(0, track, step, word);
          for (int s = 1; s < world; ++s) want = want + value_of(s, track, step, word);
          if (!ok || total != want)
          {
            if (failures.fetch_add(1) < 5)
              std::fprintf(stderr, "rank %d track %d (sequence %u) step %d word %d: got %.9g want %.9g%s\n", rank, track, sequence, step,
                  word, (double)total, (double)want, ok ? "" : " (timed out)");
          }
        }
      }
      if (doomed)
      {
        // every rank of a doomed Track leaves it (the quitter at once, the others by giving up or — if they were
        // past the quit step before the quitter got there — cannot be: the quitter's words of that step never come)
        if (!left && failures.fetch_add(1) < 5) std::fprintf(stderr, "rank %d finished a Track that another rank left\n", rank);
        if (rank == 0) aborted_tracks.fetch_add(1);
        meet(meetings++);
      }
    }
  };

  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) threads.emplace_back(rank_main, r);
  for (auto& t : threads) t.join();
  std::printf("rig_rehearsal: world %d, %d tracks x %d steps x %d words from sequence %u, %d aborted, %d failure(s)\n", world, tracks,
      steps, VK_RIG_VALUES, start_sequence, aborted_tracks.load(), failures.load());
  return failures.load() ? 1 : 0;
}
