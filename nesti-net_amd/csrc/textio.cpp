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

extern "C" {

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
