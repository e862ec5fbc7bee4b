// Host-side text output of the file seam (no GPU code): the reference writes <shape>.normals / .experts /
// .experts_probs with np.savetxt ('%.18e' / '%i', test_n_est_w_experts.py:182-188).  At ~65 k normals/s that Python
// loop costs more than the inference itself, so it is restated here on snprintf, which rounds exactly like Python's
// '%' formatting (byte-identical output is pinned by tests/test_textio.py).  Reading stays np.loadtxt + .npy cache.
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nesti_hip.h"

namespace nesti { void set_error(const std::string& msg); }

static int fail(const std::string& m) { nesti::set_error(m); return 1; }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slice-by-8 tables built once
static uint32_t g_crc_tab[8][256];
static void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
}

extern "C" {

// The checksum TensorFlow's tensor bundle stores per tensor and per table block (tensorflow/core/lib/hash/crc32c.h;
// what tf.train.Saver().restore verifies, test_n_est_w_experts.py:98-105): crc32c of `n` bytes continuing from `crc`
// (0 to start).  The stored form is masked: ((crc >> 15) | (crc << 17)) + 0xa282ead8.
uint32_t nesti_crc32c(const void* data, size_t n, uint32_t crc) {
  static const bool once = (crc_init(), true);
  (void)once;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c = g_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = g_crc_tab[7][w & 0xff] ^ g_crc_tab[6][(w >> 8) & 0xff] ^ g_crc_tab[5][(w >> 16) & 0xff] ^ g_crc_tab[4][(w >> 24) & 0xff] ^
        g_crc_tab[3][(w >> 32) & 0xff] ^ g_crc_tab[2][(w >> 40) & 0xff] ^ g_crc_tab[1][(w >> 48) & 0xff] ^ g_crc_tab[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return ~c;
}

// np.savetxt(path, float32_array.astype(float64)) with the default fmt '%.18e' and ' ' delimiter.  Rows are
// formatted by a few host threads into per-chunk buffers, then written in order.
int nesti_write_text_f32(const char* path, const float* data, int64_t rows, int cols) {
  if (!path || (!data && rows * cols > 0)) return fail("nesti_write_text_f32: null argument");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(std::string("nesti_write_text_f32: cannot open ") + path);
  unsigned hw = std::thread::hardware_concurrency();
  const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 16), rows / 4096));
  std::vector<std::vector<char>> chunks(n_thr);
  auto work = [&](int t) {
    const int64_t r0 = rows * t / n_thr, r1 = rows * (t + 1) / n_thr;
    std::vector<char>& out = chunks[t];
    out.resize((size_t)(r1 - r0) * ((size_t)cols * 26 + 1) + 32);   // '%.18e' is at most 25 chars
    size_t n = 0;
    for (int64_t r = r0; r < r1; ++r) {
      for (int c = 0; c < cols; ++c) {
        if (c) out[n++] = ' ';
        n += (size_t)snprintf(out.data() + n, 32, "%.18e", (double)data[r * cols + c]);
      }
      out[n++] = '\n';
    }
    out.resize(n);
  };
  std::vector<std::thread> thr;
  for (int t = 1; t < n_thr; ++t) thr.emplace_back(work, t);
  work(0);
  for (auto& th : thr) th.join();
  bool ok = true;
  for (int t = 0; t < n_thr; ++t) ok &= fwrite(chunks[t].data(), 1, chunks[t].size(), f) == chunks[t].size();
  fclose(f);
  return ok ? 0 : fail("nesti_write_text_f32: short write");
}

// np.savetxt(path, int_array, fmt='%i'): one value per line.
int nesti_write_text_i32(const char* path, const int32_t* data, int64_t rows) {
  if (!path || (!data && rows > 0)) return fail("nesti_write_text_i32: null argument");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(std::string("nesti_write_text_i32: cannot open ") + path);
  for (int64_t r = 0; r < rows; ++r) fprintf(f, "%i\n", data[r]);
  fclose(f);
  return 0;
}

}  // extern "C"
