// Host-side text I/O of the file seam (no GPU code): the reference parses <shape>.xyz with
// np.loadtxt(...).astype('float32') (utils/pcpnet_dataset.py:250) and writes <shape>.normals /
// .experts / .experts_probs with np.savetxt ('%.18e' / '%i', test_n_est_w_experts.py:182-188).  At ~56 k
// normals/s those Python loops cost as much as the inference itself, so they are restated here on strtod /
// snprintf, which round exactly like numpy's parser and Python's '%' formatting (byte-identical output is
// pinned by tests/test_textio.py).
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

static bool slurp(const char* path, std::vector<char>* buf) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf->resize((size_t)n + 1);
  const size_t got = fread(buf->data(), 1, (size_t)n, f);
  fclose(f);
  (*buf)[got] = 0;
  buf->resize(got + 1);
  return true;
}

extern "C" {

// Parse a whitespace-separated numeric text file.  Lines that are empty or start with '#' are skipped (like
// np.loadtxt).  Call with out == NULL to get the shape; then with a [rows, take_cols] float32 buffer to fill
// the first take_cols columns (value = (float)strtod(...), i.e. float64 parse then cast, like .astype).
int nesti_read_text_matrix(const char* path, float* out, int64_t cap_rows, int take_cols, int64_t* n_rows, int* n_cols) {
  if (!path || !n_rows || !n_cols) return fail("nesti_read_text_matrix: null argument");
  std::vector<char> buf;
  if (!slurp(path, &buf)) return fail(std::string("nesti_read_text_matrix: cannot open ") + path);
  char* p = buf.data();
  int64_t rows = 0;
  int cols = -1;
  while (*p) {
    char* eol = strchr(p, '\n');
    if (!eol) eol = p + strlen(p);
    const char save = *eol;
    *eol = 0;
    char* q = p;
    while (*q == ' ' || *q == '\t' || *q == '\r') ++q;
    if (*q && *q != '#') {
      int c = 0;
      while (*q) {
        char* end = nullptr;
        const double v = strtod(q, &end);
        if (end == q) { *eol = save; return fail(std::string("nesti_read_text_matrix: bad number in ") + path); }
        if (out && c < take_cols) {
          if (rows >= cap_rows) { *eol = save; return fail("nesti_read_text_matrix: buffer too small"); }
          out[rows * take_cols + c] = (float)v;
        }
        ++c;
        q = end;
        while (*q == ' ' || *q == '\t' || *q == '\r' || *q == ',') ++q;
        if (*q == '#') break;
      }
      if (cols < 0) cols = c;
      else if (c != cols) { *eol = save; return fail(std::string("nesti_read_text_matrix: ragged rows in ") + path); }
      if (out && c < take_cols) { *eol = save; return fail("nesti_read_text_matrix: fewer columns than requested"); }
      ++rows;
    }
    *eol = save;
    p = save ? eol + 1 : eol;
  }
  *n_rows = rows;
  *n_cols = cols < 0 ? 0 : cols;
  return 0;
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
