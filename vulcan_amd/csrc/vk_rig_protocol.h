// vk_rig_protocol.h — the words a rigid multi-camera rig's GPUs exchange inside the one-launch
// Gauss-Newton loop (BASELINE configs[4]; include/vk.h vk_rig_exchange, vk_icp_track_rig).
// No reference counterpart: the reference has no multi-GPU code.
//
// Every rank owns one AREA in its own memory, mapped into every peer (xGMI peer access): two
// Track parities x two step parities x VK_RIG_MAX_RANKS senders x VK_RIG_WORDS words of 64 bits
// {tag, value}. After a rank
// has summed its own view's 27 sums (the exchange among its workgroups, vk_gauss_newton.hpp),
// ONE workgroup writes them into every rank's area — slot [step parity][its own rank] — with
// system-scope stores that travel over the links by themselves; every workgroup of every rank
// then reads the `world` slots of its OWN area until all carry this step's tag, adds them in rank
// order (the same bits on every rank) and solves. Nothing runs on the host, no kernel ends, and
// a step costs one more trip over the fabric instead of an all-reduce and two launch boundaries.
//
// Tag = sequence (22 bits, the same on all ranks for one Track, never 0) << 10 | step + 1; value
// and tag are one atomic object, so a reader can never pair a value with the wrong step.
// Within a Track, double buffering by step parity is enough: a rank writes step i + 2 over step i
// only after it has read every rank's step i + 1, which they wrote after reading all of step i.
// Across Tracks the loops end at a step whose parity nobody knows in advance (tracker.cpp:162), so
// the buffers are doubled once more by the parity of the sequence number: a rank writes into Track
// T + 1's buffers only after it has read every rank's last step of Track T, which they wrote after
// they had finished Track T - 1 — the last user of those buffers. (The rehearsal with one step per
// Track deadlocked without this.)
//
// Plain C++ (host and device): tests/test_rig_protocol.py drives the same functions from host
// threads standing in for the ranks' workgroups.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define VK_RIG_MAX_RANKS 8
#define VK_RIG_WORDS 32          /* 27 used: gradient[6] | packed hessian[21] */
#define VK_RIG_VALUES 27

#if defined(__HIPCC__)
#define VK_RIG_FN __host__ __device__ inline
#else
#define VK_RIG_FN inline
#endif

// The sequence number after `sequence`: every rank takes this step after EVERY Track it has entered, whether the Track
// ended with a pose or with VK_TRACK_ABORTED (a retry under the old number would meet the aborted attempt's words,
// whose tags it shares). 1, 2, ... VK_RIG_LAST_SEQUENCE, 1, ...: the last number is even, so the numbers' parity — which
// picks the buffers, see above — alternates across the wrap as well; 0 is never used (a zeroed area carries no tag).
#define VK_RIG_LAST_SEQUENCE ((1u << 22) - 2u)
VK_RIG_FN uint32_t rig_next_sequence(uint32_t sequence) { return sequence >= VK_RIG_LAST_SEQUENCE ? 1u : sequence + 1u; }

VK_RIG_FN size_t rig_area_words() { return (size_t)4 * VK_RIG_MAX_RANKS * VK_RIG_WORDS; }

VK_RIG_FN size_t rig_word_index(uint32_t sequence, int step, int sender, int word)
{
  return ((size_t)((sequence & 1u) * 2u + (uint32_t)(step & 1)) * VK_RIG_MAX_RANKS + (size_t)sender) * VK_RIG_WORDS + (size_t)word;
}

VK_RIG_FN uint32_t rig_tag(uint32_t sequence, int step) { return (sequence << 10) | (uint32_t)(step + 1); }

VK_RIG_FN unsigned long long rig_pack(uint32_t tag, float value)
{
  uint32_t bits;
  memcpy(&bits, &value, 4);
  return ((unsigned long long)tag << 32) | (unsigned long long)bits;
}

VK_RIG_FN uint32_t rig_word_tag(unsigned long long word) { return (uint32_t)(word >> 32); }

VK_RIG_FN float rig_word_value(unsigned long long word)
{
  const uint32_t bits = (uint32_t)word;
  float value;
  memcpy(&value, &bits, 4);
  return value;
}

// Publish: Store(area pointer of receiver r, index, word) for every receiver.
template <typename Store>
VK_RIG_FN void rig_publish(unsigned long long* const* areas, int rank, int world, uint32_t sequence, int step, int word,
    float value, Store store)
{
  const unsigned long long packed = rig_pack(rig_tag(sequence, step), value);
  for (int r = 0; r < world; ++r) store(areas[r] + rig_word_index(sequence, step, rank, word), packed);
}

// Gather: the sum over ranks of word `word`, in rank order (the first term is taken as it is, so
// that a rig of one returns its own bits). Load(pointer) -> word; GiveUp() -> true to stop waiting.
// Returns false when a sender's word did not arrive.
template <typename Load, typename GiveUp>
VK_RIG_FN bool rig_gather(const unsigned long long* own_area, int world, uint32_t sequence, int step, int word,
    float& total, Load load, GiveUp give_up)
{
  const uint32_t tag = rig_tag(sequence, step);
  for (int s = 0; s < world; ++s)
  {
    const unsigned long long* at = own_area + rig_word_index(sequence, step, s, word);
    unsigned long long w = load(at);
    while (rig_word_tag(w) != tag)
    {
      if (give_up()) return false;
      w = load(at);
    }
    total = (s == 0) ? rig_word_value(w) : total + rig_word_value(w);
  }
  return true;
}
