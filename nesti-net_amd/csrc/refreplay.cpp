// Host-side replay of the reference's subsample stream (no GPU code).
//
// The reference thins a ball that holds n > P points with  rng.choice(n, P, replace=False)  where rng is ONE
// numpy.random.RandomState(seed) shared by every patch and scale of every shape, in visiting order
// (utils/pcpnet_dataset.py:237-240, 320-321).  numpy's legacy RandomState is frozen: choice without replacement and without
// weights is  permutation(n)[:P],  permutation is  arange(n)  shuffled by the legacy Fisher-Yates loop
//     for i in reversed(range(1, n)):  j = interval(i);  swap(x[i], x[j])
// and interval(max) draws 32-bit MT19937 outputs masked to the smallest 2^k - 1 >= max until one is <= max (rk_interval).
// The number of draws depends on the rejections, so the stream is inherently sequential -- but it only needs the ball SIZES,
// which the GPU counts in one pass (patches.hip: patches_count_kernel).  This file turns the sizes of a batch of patches into
// the pick table the reference-order patch kernel applies to the balls it has sorted into cKDTree's visiting order
// (patches.hip: patches_ref_kernel): ~1 ns per MT19937 output, 2-3 ns per shuffle step, i.e. a few tenths of a second per
// 100k-point cloud instead of the 9 s of the scipy + numpy host path (refsample.py), and it overlaps with the GPU.
// Pinned bit for bit against numpy itself by tests/test_refreplay.py.
#include <stdint.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/nesti_hip.h"

namespace nesti { void set_error(const std::string& msg); }

struct nesti_refstream {
  uint32_t mt[624];
  int pos;
  std::vector<uint32_t> perm;
};

namespace {

void mt_seed(nesti_refstream* s, uint32_t seed) {          // init_genrand: RandomState(int) -> _legacy_seeding -> mt19937_seed
  s->mt[0] = seed;
  for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->pos = 624;
}

void mt_refill(nesti_refstream* s) {
  uint32_t* mt = s->mt;
  auto twist = [](uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
  int k = 0;
  for (; k < 624 - 397; ++k) mt[k] = mt[k + 397] ^ twist(mt[k], mt[k + 1]);
  for (; k < 623; ++k) mt[k] = mt[k + 397 - 624] ^ twist(mt[k], mt[k + 1]);
  mt[623] = mt[396] ^ twist(mt[623], mt[0]);
  s->pos = 0;
}

inline uint32_t mt_next(nesti_refstream* s) {
  if (s->pos == 624) mt_refill(s);
  uint32_t y = s->mt[s->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

}  // namespace

extern "C" {

nesti_refstream_t* nesti_refstream_create(uint32_t seed) {
  nesti_refstream* s = new (std::nothrow) nesti_refstream;
  if (!s) { nesti::set_error("nesti_refstream_create: out of memory"); return nullptr; }
  mt_seed(s, seed);
  return s;
}

void nesti_refstream_destroy(nesti_refstream_t* s) { delete s; }

int nesti_refstream_picks(nesti_refstream_t* s, const int32_t* sizes, int64_t n_balls, int P, uint16_t* picks_out,
                          int64_t picks_capacity, int64_t* offsets_out, int64_t* n_over_out) {
  if (!s || (!sizes && n_balls > 0) || !offsets_out || P < 1) { nesti::set_error("nesti_refstream_picks: bad argument"); return 1; }
  int64_t used = 0, over = 0;
  // validate before drawing anything: a failed call must leave the stream where it was
  for (int64_t b = 0; b < n_balls; ++b) {
    if (sizes[b] < 0) { nesti::set_error("nesti_refstream_picks: negative ball size"); return 1; }
    if (sizes[b] > P) {
      if (sizes[b] > 65535) { nesti::set_error("nesti_refstream_picks: a ball holds more than 65535 points (uint16 pick table)"); return 1; }
      ++over;
    }
  }
  if (over * (int64_t)P > picks_capacity || (over && !picks_out)) { nesti::set_error("nesti_refstream_picks: pick table too small"); return 1; }
  for (int64_t b = 0; b < n_balls; ++b) {
    const uint32_t n = (uint32_t)sizes[b];
    if (n <= (uint32_t)P) { offsets_out[b] = -1; continue; }          // utils/pcpnet_dataset.py:320: only over-full balls draw
    if (s->perm.size() < n) s->perm.resize(n);
    uint32_t* x = s->perm.data();
    for (uint32_t i = 0; i < n; ++i) x[i] = i;
    uint32_t mask = n - 1;                                            // smallest 2^k - 1 >= i, kept incrementally as i falls
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (uint32_t i = n - 1; i >= 1; --i) {
      while ((mask >> 1) >= i) mask >>= 1;
      uint32_t j;
      do { j = mt_next(s) & mask; } while (j > i);
      const uint32_t t = x[i]; x[i] = x[j]; x[j] = t;
    }
    uint16_t* dst = picks_out + used;
    for (int r = 0; r < P; ++r) dst[r] = (uint16_t)x[r];
    offsets_out[b] = used;
    used += P;
  }
  if (n_over_out) *n_over_out = over;
  return 0;
}

}  // extern "C"
