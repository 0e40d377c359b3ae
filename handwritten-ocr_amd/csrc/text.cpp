// Native integer kernels for the candidate compare / merge step of a page read (host C++, libhwocr_text.so).
//
// The reference does these in pure CPython on the hot path of node_initial_ocr:
//   levenshtein / _levenshtein_words   ocr_agent/tools.py:69-100   (single-row DP, O(n*m) interpreter steps)
//   _align_to_backbone                 ocr_agent/tools.py:465-493  ((n+1)x(m+1) LCS table of Python lists)
// Results are integers and must be bit-exact, including the tie-breaking of the LCS backtrack.
#include "hwocr.h"

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

// Myers / Hyyro bit-parallel edit distance, 64 pattern rows per machine word, blocks chained by the
// horizontal delta (-1, 0, +1) that leaves the bottom row of each block.
int64_t myers_blocks(const uint32_t* pat, int64_t m, const uint32_t* txt, int64_t n) {
  const int64_t nb = (m + 63) / 64;
  std::unordered_map<uint32_t, int64_t> slot;  // symbol -> row of peq
  slot.reserve((size_t)m * 2);
  std::vector<uint64_t> peq;
  for (int64_t i = 0; i < m; ++i) {
    auto it = slot.find(pat[i]);
    int64_t s;
    if (it == slot.end()) {
      s = (int64_t)slot.size();
      slot.emplace(pat[i], s);
      peq.resize((size_t)(s + 1) * nb, 0);
    } else {
      s = it->second;
    }
    peq[(size_t)s * nb + i / 64] |= 1ULL << (i % 64);
  }
  std::vector<uint64_t> pv((size_t)nb, ~0ULL), mv((size_t)nb, 0ULL);
  const std::vector<uint64_t> zeros((size_t)nb, 0ULL);
  const uint64_t last_bit = 1ULL << ((m - 1) % 64);
  int64_t score = m;
  for (int64_t j = 0; j < n; ++j) {
    auto it = slot.find(txt[j]);
    const uint64_t* eqrow = it == slot.end() ? zeros.data() : &peq[(size_t)it->second * nb];
    int hin = 1;  // top boundary D[0][j] grows by one per column
    for (int64_t b = 0; b < nb; ++b) {
      uint64_t eq = eqrow[b];
      const uint64_t pvb = pv[b], mvb = mv[b];
      const uint64_t xv = eq | mvb;
      if (hin < 0) eq |= 1ULL;
      const uint64_t xh = (((eq & pvb) + pvb) ^ pvb) | eq;
      uint64_t ph = mvb | ~(xh | pvb);
      uint64_t mh = pvb & xh;
      const uint64_t top = (b == nb - 1) ? last_bit : (1ULL << 63);
      int hout = 0;
      if (ph & top) hout = 1;
      else if (mh & top) hout = -1;
      ph <<= 1;
      mh <<= 1;
      if (hin < 0) mh |= 1ULL;
      else if (hin > 0) ph |= 1ULL;
      pv[b] = mh | ~(xv | ph);
      mv[b] = ph & xv;
      hin = hout;
    }
    score += hin;
  }
  return score;
}

}  // namespace

extern "C" int64_t hwocr_levenshtein_u32(const uint32_t* a, int64_t n, const uint32_t* b, int64_t m) {
  if (n < 0 || m < 0) return -1;
  if (n == 0) return m;
  if (m == 0) return n;
  // pattern = the shorter string (fewer blocks)
  if (n <= m) return myers_blocks(a, n, b, m);
  return myers_blocks(b, m, a, n);
}

extern "C" int hwocr_lcs_align_u32(const uint32_t* backbone, int64_t n, const uint32_t* words, int64_t m,
                                   int32_t* out) {
  if (n < 0 || m < 0 || (n > 0 && !out)) return HWOCR_EINVAL;
  for (int64_t i = 0; i < n; ++i) out[i] = -1;
  if (n == 0 || m == 0) return HWOCR_OK;
  const int64_t W = m + 1;
  std::vector<int32_t> dp((size_t)(n + 1) * W, 0);
  for (int64_t i = 1; i <= n; ++i) {
    const uint32_t bi = backbone[i - 1];
    const int32_t* up = &dp[(size_t)(i - 1) * W];
    int32_t* cur = &dp[(size_t)i * W];
    for (int64_t j = 1; j <= m; ++j)
      cur[j] = (bi == words[j - 1]) ? up[j - 1] + 1 : std::max(up[j], cur[j - 1]);
  }
  int64_t i = n, j = m;
  while (i > 0 && j > 0) {
    if (backbone[i - 1] == words[j - 1]) {
      out[i - 1] = (int32_t)(j - 1);
      --i; --j;
    } else if (dp[(size_t)(i - 1) * W + j] >= dp[(size_t)i * W + (j - 1)]) {
      --i;
    } else {
      --j;
    }
  }
  return HWOCR_OK;
}
